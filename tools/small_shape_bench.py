"""ms per forward of the small / mid batch shapes (status mode 'lazy', as the drivers run them) and the layer-0 projection's share (library events):
    python tools/small_shape_bench.py [B,T ...]       (A/B: TEPOSE_AMD_LIB=<other build>, kernel knobs through the environment)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(64, 16), (37, 6), (128, 16), (256, 16), (512, 6)]
dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np)
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
eng = model._engine
for B, T in shapes:
    x = synthetic_windows_device(B, T, 77, dev)
    with torch.no_grad(), eng.lazy_status():
        for _ in range(5):
            out = model(x, J_regressor=J)[0]
        best = 1e9
        for rnd in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                model(x, J_regressor=J)
            eng.check_status()
            best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
        eng.profile_enable(True)
        for _ in range(20):
            model(x, J_regressor=J)
        torch.cuda.synchronize()
        k_ms, k_n, _ = eng.profile_read()
        eng.profile_enable(False)
    print('B=%d T=%d: %.4f ms per forward | layer-0 projection %.4f ms (%s) | verts checksum %.6f'
          % (B, T, best, k_ms / max(k_n, 1), eng.select_kernels(B, T).get('projection'), float(out['verts'].double().abs().sum())), flush=True)
