"""Time the library's fp32-MFMA GEMM alone (tepose_gemm_f32) on random vs zero operands.
Zero data draws less power, so a large gap means the kernel is clock(DVFS)-limited rather
than schedule-limited (MI355X_MICROARCH.md, DVFS give-back)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import _lib  # noqa: E402


def run(M, N, K, fill, reps=6):
    lib = _lib.load()
    dev = torch.device('cuda')
    if fill == 'zero':
        A = torch.zeros(M, K, device=dev); W = torch.zeros(N, K, device=dev)
    else:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(lib.tepose_gemm_workspace_bytes(N, K), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = lib.tepose_gemm_f32(A.data_ptr(), K, W.data_ptr(), K, None, C.data_ptr(), N, M, N, K, 0,
                                 ws.data_ptr(), ws.numel(), st)
        assert rc == 0
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('M=%d N=%d K=%d %-6s %8.3f ms  %6.1f TFLOP/s (incl. %d-row weight pack)' %
          (M, N, K, fill, ms, 2.0 * M * N * K / ms / 1e9, N), flush=True)


if __name__ == '__main__':
    for fill in ('random', 'zero', 'random'):
        run(65536, 9216, 2144, fill)
    run(65536, 3072, 1024, 'random')
    run(65536, 3072, 2048, 'random')
