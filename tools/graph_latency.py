"""Eager forward vs the same forward captured into a hipGraph, B = 1: host cost per forward and end-to-end time."""
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
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np)
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
for B, T in ((1, 6), (1, 16), (1, 32)):
    x = synthetic_windows_device(B, T, 7, dev)
    with torch.no_grad():
        for _ in range(5):
            model(x, J_regressor=J)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(x, J_regressor=J)[0]
        res = {}
        for name, f in (('eager', lambda: model(x, J_regressor=J)), ('graph replay', g.replay)):
            for _ in range(10):
                f()
            torch.cuda.synchronize()
            host = 0.0
            for _ in range(30):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    f()
                host += (time.perf_counter() - t0) / 4
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                f()
            torch.cuda.synchronize()
            res[name] = (host / 30 * 1e6, (time.perf_counter() - t0) / 300 * 1e6)
    print('B=%d T=%2d  ' % (B, T) + '  |  '.join('%s: host %.1f us, end-to-end %.1f us' % ((k,) + v) for k, v in res.items()), flush=True)
