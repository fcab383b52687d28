"""Experiment: does a cfg-C step get cheaper as two half-batches on two streams (the HBM-bound kernels of one half -- input split,
first cell steps, blend shapes, skinning -- under the MFMA-bound kernels of the other)?  Two handles share one packed blob.
    python tools/two_stream_halves.py [B] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
a, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np)
b, _, _ = build_model(2, 1024, seed=1, device=dev, smpl_np=smpl_np)
x = synthetic_windows_device(B, 16, 3, dev)
with torch.no_grad():
    a(x[:4], J_regressor=J)
    b._engine.adopt_blob(a._engine.blob, b)
    xs = [x[:B // 2].contiguous(), x[B // 2:].contiguous()]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def full():
        return a(x, J_regressor=J)[0]

    def halves():
        with torch.cuda.stream(s1):
            o1 = a(xs[0], J_regressor=J)[0]
        with torch.cuda.stream(s2):
            o2 = b(xs[1], J_regressor=J)[0]
        return o1, o2

    for name, f in (('one batch, one stream', full), ('two halves, two streams', halves), ('one batch, one stream', full),
                    ('two halves, two streams', halves)):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print('%-26s %8.3f ms per %d windows = %9.0f windows/s' % (name, dt * 1e3, B, B / dt), flush=True)
