"""Soak test of the persistent kernels' hand-offs: thousands of forwards per batch size, each compared bitwise with the
first, while a second stream (and optionally a second process running this same script) loads the GPU unevenly."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np)
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
side = torch.cuda.Stream()
junk = torch.randn(48 << 20, device=dev)
bad = {}
t0 = time.time()
with torch.no_grad():
    for B, T in ((1, 16), (1, 32), (2, 7), (3, 16), (4, 5), (5, 16), (8, 9), (16, 16), (17, 6), (32, 16), (33, 5), (48, 16), (64, 16), (64, 3)):
        x = synthetic_windows_device(B, T, 100 + B, dev)
        ref = {k: v.clone() for k, v in model(x, J_regressor=J)[0].items()}
        n_bad = 0
        for it in range(iters):
            if it % 7 == 0:
                with torch.cuda.stream(side):
                    junk.mul_(1.0001).add_(0.25)
                    if it % 21 == 0:
                        (junk[:1 << 22].view(2048, 2048) @ junk[1 << 22:1 << 23].view(2048, 2048)).sum()
            out = model(x, J_regressor=J)[0]
            if it % 4 == 3 and any(not torch.equal(out[k], ref[k]) for k in out):
                n_bad += 1
        bad['%dx%d' % (B, T)] = n_bad
        print('B=%2d T=%2d: %d mismatching of %d checks  (%.0f s)' % (B, T, n_bad, iters // 4, time.time() - t0), flush=True)
print('soak mismatches', bad)
