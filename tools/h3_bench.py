"""Split-precision (fp16x3) GEMM prototype: accuracy vs fp64 and time vs the exact-fp32 kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)


def run(M, N, K, check):
    A = torch.randn(M, K, device=dev, generator=g).abs() * 0.5
    W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.03
    b = torch.randn(N, device=dev, generator=g)
    C3 = torch.full((M, N), float('nan'), device=dev)
    C1 = torch.empty(M, N, device=dev)
    ws3 = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)
    ws1 = torch.empty(lib.tepose_gemm_workspace_bytes(N, K), dtype=torch.uint8, device=dev)

    def h3():
        assert lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C3.data_ptr(), N, M, N, K,
                                      ws3.data_ptr(), ws3.numel(), st) == 0

    def f32():
        assert lib.tepose_gemm_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C1.data_ptr(), N, M, N, K, 0,
                                   ws1.data_ptr(), ws1.numel(), st) == 0
    h3(); f32(); torch.cuda.synchronize()
    if check:
        ref = A.double() @ W.double().t() + b.double()
        e3 = (C3.double() - ref).abs().max().item()
        e1 = (C1.double() - ref).abs().max().item()
        print('M=%d N=%d K=%d  max|err| split %.2e  fp32 %.2e  (|C| max %.1f)' % (M, N, K, e3, e1, ref.abs().max().item()))
    for name, fn in (('split fp16x3', h3), ('exact fp32  ', f32)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print('   %s %8.3f ms  %7.1f TFLOP/s (incl. operand split/pack passes)' % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)


run(300, 200, 96, True)
run(4096, 3072, 1024, True)
run(65536, 9216, 2144, False)
run(65536, 3072, 1024, False)
