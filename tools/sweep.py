"""ms per TePose.forward and windows/s for a sweep of (B, T) on one GPU (not the bench line;
used to see the small-batch / latency regime of BASELINE.json configs 1, 2 and 5)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402


def main():
    shapes = [(1, 16), (1, 32), (1, 6), (8, 16), (64, 16), (256, 16), (1024, 16), (4096, 16)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
    dev = torch.device('cuda', 0)
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    for B, T in shapes:
        x = synthetic_windows_device(B, T, 7, dev)
        # forwards queued back to back: status mode 'lazy', the persistent kernels' fault word is read once at the end
        with torch.no_grad(), model._engine.lazy_status():
            for _ in range(3):
                model(x, J_regressor=J)
            torch.cuda.synchronize()
            n = max(3, min(200, int(2000 / max(1, B // 8))))
            t0 = time.perf_counter()
            for _ in range(n):
                model(x, J_regressor=J)
            model._engine.check_status()
            dt = (time.perf_counter() - t0) / n
        print('B=%5d T=%2d  %9.3f ms/forward  %10.1f windows/s' % (B, T, dt * 1e3, B / dt), flush=True)


if __name__ == '__main__':
    main()
