import os, sys, torch
sys.path.insert(0, os.getcwd())
from tepose_amd import _lib
lib = _lib.load()
os.environ['TEPOSE_H3S'] = sys.argv[1]
g = torch.Generator(device='cuda').manual_seed(9)
st = torch.cuda.current_stream().cuda_stream
for M, N, K, ldc, use_bias in [(8192, 2304, 2144, 2304, True), (5000, 5000, 256, 5000, True), (4096, 4608, 64, 4608, False), (2100, 2050, 512, 2051, True),
                               (300, 200, 128, 200, False), (256, 256, 1024, 256, True), (65536, 512, 96, 512, True), (1500, 1024, 192, 1024, True), (777, 333, 352, 340, True)]:
    A = torch.randn(M, K, device='cuda', generator=g) * 3.0
    W = torch.randn(N, K, device='cuda', generator=g) * 0.05
    b = torch.randn(N, device='cuda', generator=g) if use_bias else None
    C = torch.full((M, ldc), float('nan'), device='cuda')
    ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
    rc = lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr() if use_bias else None, C.data_ptr(), ldc, M, N, K, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t()
    if use_bias: ref += b.double()
    mag = (A.double().abs() @ W.double().abs().t()).max().item() + 1.0
    err = (C[:, :N].double() - ref).abs().max().item()
    print(M, N, K, 'rc', rc, 'err/mag %.2e' % (err / mag), 'ok' if err < 3e-6 * mag and torch.isnan(C[:, N:]).all() else 'FAIL', flush=True)
