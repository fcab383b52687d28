"""(Needs the experiment wired in: see the header of tools/micro/gru_step16w_experiment.hip; record: profiles/r06_gru_bigtile.txt.)
Same-box A/B of the fused GRU step: gru_step16_kernel<true> (128 x 192 tiles, two workgroups per CU, barriers) against gru_step16w_kernel (256 x 192
tiles on the barrier-free pipeline, one persistent workgroup per CU; TEPOSE_GRU_WIDE=1).  Per batch: bit-identity of the encoder features, then the
recurrent part per forward (hipEvents around every layer's step sequence: tepose_profile_read_gru) and the whole encoder, interleaved A B A B.

    python tools/gru_wide_ab.py [B ...]      (default 8192 2048)
    TEPOSE_GRUW_GM=4 python tools/gru_wide_ab.py 8192
"""
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
state = synth.synthetic_state_dict(2, 1024, 0)
base, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np, state=state)
os.environ['TEPOSE_GRU_WIDE'] = '1'
wide, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np, state=state)
del os.environ['TEPOSE_GRU_WIDE']
T = 16
for B in [int(a) for a in sys.argv[1:]] or [8192, 2048]:
    x = synthetic_windows_device(B, T, 3, dev)
    print('B=%d  base: %s | wide: %s' % (B, base._engine.select_kernels(B, T)['gru_step_l1'], wide._engine.select_kernels(B, T)['gru_step_l1']))
    with torch.no_grad():
        fa, fb = base.encoder(x), wide.encoder(x)
        torch.cuda.synchronize()
        print('  features bit-identical: %s  (max |diff| %.3e, finite %s)' % (bool(torch.equal(fa, fb)), float((fa - fb).abs().max()), bool(torch.isfinite(fb).all())))
        res = {'base': [], 'wide': []}
        for rep in range(3):
            for name, mdl in (('base', base), ('wide', wide)):
                eng = mdl._engine
                mdl.encoder(x)
                torch.cuda.synchronize()
                eng.profile_enable(True)
                t0 = time.perf_counter()
                for _ in range(4):
                    mdl.encoder(x)
                torch.cuda.synchronize()
                enc = (time.perf_counter() - t0) / 4 * 1e3
                g_ms, g_n, g_fl = eng.profile_read_gru()
                eng.profile_enable(False)
                res[name].append((g_ms / g_n, enc, g_fl / (g_ms / g_n * 1e-3) / 1e12 / (2516.6 / 3)))
        for name in ('base', 'wide'):
            print('  %s: ' % name + '  '.join('gru %.3f ms (frac %.3f) enc %.2f ms' % (g, f, e) for g, e, f in res[name]))
        b, w = min(r[0] for r in res['base']), min(r[0] for r in res['wide'])
        print('  recurrent part: base %.3f ms, wide %.3f ms (%+.1f %%)' % (b, w, (w / b - 1) * 100))
