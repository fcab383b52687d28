"""Time the encoder alone (tepose_encoder_fwd) at a given batch; used with ablation builds."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda', 0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=synth.synthetic_smpl(0))
x = synthetic_windows_device(B, 16, 3, dev)
with torch.no_grad():
    for _ in range(2):
        model.encoder(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        model.encoder(x)
    torch.cuda.synchronize()
print('encoder B=%d: %.2f ms' % (B, (time.perf_counter() - t0) / 4 * 1e3))
