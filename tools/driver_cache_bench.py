"""Window-driver throughput with and without the layer-0 projection cache."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import synth  # noqa: E402
from tepose_amd.driver import run_clips  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

dev = torch.device('cuda', 0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=synth.synthetic_smpl(0))
g = torch.Generator(device=dev).manual_seed(1)
for C, n, T in ((37, 300, 16), (256, 80, 16), (1024, 48, 16), (4096, 40, 16), (1024, 30, 6)):
    feats = [torch.randn(n, 2048, device=dev, generator=g).abs() * 0.5 for _ in range(C)]
    inits = [torch.randn(T - 1, 85, device=dev, generator=g) * 0.2 for _ in range(C)]
    for cache in (False, True):
        run_clips(model, feats[:2], inits[:2], T, keep=('theta',), cache_projections=cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_clips(model, feats, inits, T, keep=('theta', 'kp_3d'), cache_projections=cache)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        w = C * (n - T + 1)
        print('clips=%4d frames=%3d T=%2d cache=%-5s %7.3f s  %9.0f windows/s' % (C, n, T, cache, dt, w / dt), flush=True)
