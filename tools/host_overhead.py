"""Host-side cost of one forward (issue rate without waiting for the GPU) next to the GPU-bound time, B = 1."""
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
        for _ in range(20):
            model(x, J_regressor=J)
        torch.cuda.synchronize()
        # host cost: a few forwards issued into an EMPTY queue (nothing to wait for), repeated
        t_issue = 0.0
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                model(x, J_regressor=J)
            t_issue += (time.perf_counter() - t0) / 4
        t_issue /= 30
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            model(x, J_regressor=J)
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / n
    print('B=%d T=%2d  host issue %.1f us per forward, end-to-end %.1f us per forward' % (B, T, t_issue * 1e6, t_all * 1e6), flush=True)
