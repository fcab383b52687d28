"""Per-frame latency of the live-stream path (BASELINE.json config 5): one clip, sliding window of
`seqlen` frames advancing one frame at a time with theta feedback (demo.py:238-252)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import synth  # noqa: E402
from tepose_amd.driver import run_clips  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np)
for T in (32, 16, 6):
    n = 400 + T
    w = synth.synthetic_windows(1, n, 5)[0]
    feats = [torch.from_numpy(w[:, :2048].copy()).to(dev)]
    init = [torch.from_numpy(w[:T - 1, 2048:].copy()).to(dev)]
    run_clips(model, feats, init, T, keep=('theta', 'verts', 'kp_3d'))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_clips(model, feats, init, T, keep=('theta', 'verts', 'kp_3d'))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('seqlen=%2d: %d frames in %.3f s -> %.3f ms per frame (%.0f fps)' % (T, n - T + 1, dt, dt / (n - T + 1) * 1e3,
                                                                              (n - T + 1) / dt), flush=True)
