"""A/B of the blend-shape product of a large batch: gemm_h3_kernel (two-accumulator planes) vs gemm_h3s_persist16c_kernel (scaled planes), whole forwards:
    python tools/blend_ab.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device
from tepose_amd import synth
from tepose_amd.testing import build_model
dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
state = synth.synthetic_state_dict(2, 1024, 0)
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
outs = {}
for name, v in (('old', str(0x7fffffff)), ('new', '512')):
    os.environ['TEPOSE_BLEND16_MIN_N'] = v
    m, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np, state=state)
    for B in (8192, 1000):
        x = synthetic_windows_device(B, 16, 3, dev)
        with torch.no_grad():
            for _ in range(2):
                o = m(x, J_regressor=J)[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                o = m(x, J_regressor=J)[0]
            torch.cuda.synchronize()
        print(name, B, 'ms/forward %.3f' % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
        outs[(name, B)] = {k: o[k].clone() for k in ('verts', 'kp_3d', 'theta')}
for B in (8192, 1000):
    for k in ('verts', 'kp_3d', 'theta'):
        print(B, k, 'max|new - old| %.3e' % float((outs[('new', B)][k] - outs[('old', B)][k]).abs().max()), 'finite', bool(torch.isfinite(outs[('new', B)][k]).all()))
