"""Products of 37 .. 1024 rows on the split-precision planes: the tile kernel (gemm_h3.hip; TEPOSE_H3_TILE64=0 / TEPOSE_H3_TILE=64 force its row tile) against the
width-first kernel (skinny_h3.hip) -- accuracy against fp64 here, kernel times from a rocprofv3 kernel trace of this script (tools/mid_rows_gemm_digest.py):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mid -- python3 tools/mid_rows_gemm_bench.py [MxNxK ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [
    (74, 9216, 2144), (128, 9216, 2144), (222, 9216, 2144), (288, 3072, 2048), (288, 3072, 1024), (444, 9216, 2144), (768, 9216, 2144), (1024, 9216, 2144),
    (1024, 3072, 2048), (37, 9216, 2144), (64, 9216, 2144), (100, 2048, 2048), (222, 157, 1024)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev, generator=g).abs() * 0.5
    W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.03
    b = torch.randn(N, device=dev, generator=g)
    ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)
    out = {}
    for mode in ('0', 'skinny'):
        os.environ['TEPOSE_H3S'] = mode
        C = torch.full((M, N), float('nan'), device=dev)
        for _ in range(4):
            rc = lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, M, N, K, ws.data_ptr(), ws.numel(), st)
            assert rc == 0, (mode, rc)
        torch.cuda.synchronize()
        out[mode] = C
    ref = A.double() @ W.double().t() + b.double()
    print('M=%d N=%d K=%d: max|err| vs fp64: tiles %.2e width-first %.2e' % (M, N, K, (out['0'].double() - ref).abs().max().item(),
                                                                          (out['skinny'].double() - ref).abs().max().item()), flush=True)
