"""Prototype check (gemm_h3s.hip, TEPOSE_H3S=1 selects it inside tepose_gemm_h3_f32): accuracy vs an fp64 product and
timing on the layer-0 projection shape, next to the shipped split kernel.  Run under
`rocprofv3 --kernel-trace --stats` for kernel-only times."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from tepose_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda')
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)

    def run(M, N, K, check, reps):
        A = torch.randn(M, K, device=dev, generator=g).abs() * 0.5
        W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.03
        C = torch.full((M, N), float('nan'), device=dev)
        ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)

        def f():
            assert lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, None, C.data_ptr(), N, M, N, K,
                                          ws.data_ptr(), ws.numel(), st) == 0
        f(); torch.cuda.synchronize()
        if check:
            ref = A.double() @ W.double().t()
            mag = (A.double().abs() @ W.double().abs().t()).max().item()
            print('   M=%d N=%d K=%d  max|err| %.2e  (3e-6 * sum|a||w| = %.2e)' % (M, N, K, (C.double() - ref).abs().max().item(), 3e-6 * mag))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        print('   M=%d N=%d K=%d  %.3f ms per call incl. the operand split passes' % (M, N, K, e0.elapsed_time(e1) / reps))
    run(1000, 700, 2144, True, 1)
    run(300, 256, 96, True, 1)
    run(131072, 9216, 2144, False, 4)
    sys.exit(0)
for name, env in (('shipped gemm_h3_kernel', {}), ('prototype gemm_h3s_kernel', {'TEPOSE_H3S': '1'})):
    print(name, flush=True)
    subprocess.check_call([sys.executable, __file__, 'child'], env=dict(os.environ, **env))
