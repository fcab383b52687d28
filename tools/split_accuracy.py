"""Accuracy of the large-batch (split-precision) path vs the exact-fp32 path, both against the fp64 oracle,
on the benchmark architecture (L=2, H=1024, T=16)."""
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from tepose_amd import synth
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    model, state, _ = build_model(2, 1024, seed=0, device='cuda', smpl_np=smpl_np)
    x = synth.synthetic_windows(1024, 16, 4242)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        out = model(torch.from_numpy(x).cuda(), J_regressor=J)[0]
    np.savez(sys.argv[2], **{k: v[:24].cpu().numpy() for k, v in out.items()})
    sys.exit(0)

from oracle import tepose_ref as O
from tepose_amd import synth
smpl_np = synth.synthetic_smpl(0)
state = synth.synthetic_state_dict(2, 1024, 0)
x = synth.synthetic_windows(1024, 16, 4242)[:24]
ref = O.tepose_fwd(state, smpl_np, x, 2, J_regressor=smpl_np['J_regressor_h36m'], dtype=torch.float64)
for name, env in (('split fp16x3 (default, B=1024)', {}), ('exact fp32 (TEPOSE_EXACT_FP32=1)', {'TEPOSE_EXACT_FP32': '1'})):
    f = '/tmp/acc_%s.npz' % ('split' if not env else 'exact')
    subprocess.check_call([sys.executable, __file__, 'child', f], env=dict(os.environ, **env))
    got = np.load(f)
    print(name)
    for k in ('verts', 'kp_3d', 'rotmat', 'kp_2d'):
        print('   %-7s max|gpu - fp64 oracle| = %.2e' % (k, np.abs(got[k] - ref[k].numpy()).max()))
    th, rt = got['theta'], ref['theta'].numpy()
    print('   theta (cam, betas)            = %.2e' % max(np.abs(th[:, :3] - rt[:, :3]).max(), np.abs(th[:, 75:] - rt[:, 75:]).max()))
