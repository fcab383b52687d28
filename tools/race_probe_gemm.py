"""Diagnostic (known issues: two processes sharing one GPU): the split-precision and fp32 GEMM test entries repeated on fixed operands, bitwise compare (run two copies)"""
import sys
import torch
sys.path.insert(0, sys.argv[1])
from tepose_amd import _lib, synth
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def mk(M, N, K):
    A = torch.from_numpy(synth.normal('rgA%d' % M, (M, K))).cuda()
    W = torch.from_numpy(synth.normal('rgW%d' % N, (N, K))).cuda()
    ldc = (N + 127) // 128 * 128 if M == 20 else N        # 20672-like padded row stride: the vectorised C stores
    C = torch.empty(M, ldc, device='cuda')
    ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
    ws2 = torch.empty(lib.tepose_gemm_workspace_bytes(N, K), dtype=torch.uint8, device='cuda')
    return A, W, C, ws, ws2
cases = {'h3_20x20670x224': (20, 20670, 224, 1), 'h3_300x3072x1024': (300, 3072, 1024, 1), 'f32_20x20670x224': (20, 20670, 224, 0),
         'h3_1100x2048x1024': (1100, 2048, 1024, 1)}
bufs = {k: mk(*v[:3]) for k, v in cases.items()}
def run(k):
    M, N, K, h3 = cases[k]
    A, W, C, ws, ws2 = bufs[k]
    C.fill_(float('nan'))
    if h3:
        rc = lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, None, C.data_ptr(), C.stride(0), M, N, K, ws.data_ptr(), ws.numel(), st)
    else:
        rc = lib.tepose_gemm_f32(A.data_ptr(), K, W.data_ptr(), K, None, C.data_ptr(), C.stride(0), M, N, K, 0, ws2.data_ptr(), ws2.numel(), st)
    assert rc == 0
    return C[:, :N]
ref = {k: run(k).clone() for k in cases}
bad = {k: 0 for k in cases}
for it in range(int(sys.argv[2])):
    for k in cases:
        C = run(k)
        if not torch.equal(C, ref[k]):
            bad[k] += 1
            ne = (C != ref[k]) | torch.isnan(C)
            rows = ne.any(1).nonzero().flatten().tolist()
            cols = ne.any(0).nonzero().flatten().tolist()
            if bad[k] <= 6:
                print('iter %d %s: %d wrong, rows %s cols %s..%s (n=%d) max %.3e' % (it, k, int(ne.sum()), rows[:24], cols[:4], cols[-2:], len(cols),
                                                                                 float((C - ref[k]).abs().nan_to_num(9e9).max())), flush=True)
print('mismatches', bad)
