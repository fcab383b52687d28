"""Run the split-precision GEMM a few times on one shape (for rocprofv3 counter passes / ablation builds
selected with TEPOSE_AMD_LIB):  python tools/h3_loop.py M N K [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import _lib  # noqa: E402

lib = _lib.load()
M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).abs() * 0.5
W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.03
C = torch.empty(M, N, device=dev)
ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)
for _ in range(reps):
    assert lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, ws.data_ptr(),
                                  ws.numel(), st) == 0
torch.cuda.synchronize()
print('done', M, N, K, reps)
