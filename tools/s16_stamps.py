"""Phase timeline of one wave of gemm_h3s_persist16_kernel (diagnostic build with -DTEPOSE_S16_STAMPS=1):
    tools/build_abl.sh gemm_h3s16 TEPOSE_S16_STAMPS s16stamps 1 ; TEPOSE_AMD_LIB=build/abl/lib_s16stamps1.so python tools/s16_stamps.py
Stamps (s_memtime, shader clock cycles) of wave 0 of workgroup 0: 1 before the vmcnt wait, 2 after it, 3 after barrier p,
4/5/6 after the lgkmcnt wait of quarter 0/1/2, 7 after the last wait, 8 after barrier B'."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import _lib  # noqa: E402

lib = _lib.load()
os.environ['TEPOSE_H3S'] = '2'
M, N, K = 131072, 9216, 2144
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).abs() * 0.5
W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.03
C = torch.empty(M, N, device=dev)
ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)
for _ in range(3):
    assert lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, 0, C.data_ptr(), N, M, N, K, ws.data_ptr(), ws.numel(), st) == 0
torch.cuda.synchronize()
n = 8192
buf = (ctypes.c_ulonglong * n)()
raw = ctypes.CDLL(os.environ.get('TEPOSE_AMD_LIB'))
raw.tepose_debug_s16_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.tepose_debug_s16_stamps(buf, n) == 0
ev = [(v >> 8, v & 255) for v in buf if v]
# steady-state pairs of the second tile: average cycles between consecutive stamps, by (from, to)
import collections
acc = collections.defaultdict(list)
start = 8 * 80            # skip the first tile's first pairs
for (t0, i0), (t1, i1) in zip(ev[start:start + 8 * 40], ev[start + 1:start + 8 * 40 + 1]):
    acc[(i0, i1)].append(t1 - t0)
tot = 0.0
for k in sorted(acc):
    v = acc[k]
    m = sum(v) / len(v)
    tot += m
    print('stamp %d -> %d: %8.1f cycles (n=%d, min %d max %d)' % (k[0], k[1], m, len(v), min(v), max(v)))
print('sum per pair step: %.0f shader cycles' % tot)
