import os, sys, time, torch
sys.path.insert(0, ".")
from tepose_amd import synth
from tepose_amd.driver import run_clips
from tepose_amd.testing import build_model
dev = torch.device('cuda', 0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=synth.synthetic_smpl(0))
g = torch.Generator(device=dev).manual_seed(1)
for T in (6, 16, 32):
    for C in (1, 2, 3, 4, 6, 8, 16):
        n = 300
        feats = [torch.randn(n, 2048, device=dev, generator=g).abs() * 0.5 for _ in range(C)]
        inits = [torch.randn(T - 1, 85, device=dev, generator=g) * 0.2 for _ in range(C)]
        r = []
        for cache in (False, True):
            run_clips(model, feats, inits, T, keep=('theta', 'kp_3d', 'verts'), cache_projections=cache)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run_clips(model, feats, inits, T, keep=('theta', 'kp_3d', 'verts'), cache_projections=cache)
            torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / (n - T + 1) * 1e3)
        print('T=%2d clips=%2d  ms per lock-step: nocache %.4f  cache %.4f  %s' % (T, C, r[0], r[1], 'cache wins' if r[1] < r[0] else ''), flush=True)
