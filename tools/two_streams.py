"""Two models issuing small-batch forwards on two streams of ONE process (persistent kernels of both in flight together):
do they complete, are results right, how long does it take?  (small batches)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
ma, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np)
mb, _, _ = build_model(2, 1024, seed=1, device=dev, smpl_np=smpl_np)
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
xa, xb = synthetic_windows_device(1, 16, 7, dev), synthetic_windows_device(3, 16, 8, dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    ra = ma(xa, J_regressor=J)[0]['verts'].clone()
    rb = mb(xb, J_regressor=J)[0]['verts'].clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bad = 0
    for it in range(200):
        with torch.cuda.stream(sa):
            oa = ma(xa, J_regressor=J)[0]['verts']
        with torch.cuda.stream(sb):
            ob = mb(xb, J_regressor=J)[0]['verts']
        if it % 20 == 19:
            torch.cuda.synchronize()
            bad += int(not torch.equal(oa, ra)) + int(not torch.equal(ob, rb))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print('200 x 2 concurrent forwards: %.3f s (%.3f ms per pair), mismatching checks: %d, finite: %s' %
      (dt, dt / 200 * 1e3, bad, bool(torch.isfinite(oa).all() and torch.isfinite(ob).all())))
