"""GPU: the collapsed regressor.  In eval mode the reference's FC loop (spin.py:252-261) has no activation
and Dropout is the identity, so three iterations from the model's own initial state are one affine map of the feature,
and through the tail linears of relu(final GRU states).  The library forms that map in fp64 at pack time and runs ONE product
where the reference runs ten dependent ones.  Checked here against the loop itself (TEPOSE_COLLAPSE_REGRESSOR=0 handles, same
weights), for every entry that can take the shortcut and every case that must not."""
import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
KEYS = ('theta', 'verts', 'kp_3d', 'kp_2d', 'rotmat')


def _pair(monkeypatch, L, H, seed, smpl_np):
    from tepose_amd.testing import build_model
    fast, state, _ = build_model(L, H, seed=seed, device='cuda', smpl_np=smpl_np)
    monkeypatch.setenv('TEPOSE_COLLAPSE_REGRESSOR', '0')
    loop, _, _ = build_model(L, H, seed=seed, device='cuda', smpl_np=smpl_np)
    monkeypatch.delenv('TEPOSE_COLLAPSE_REGRESSOR')
    return fast, loop, state


@pytest.mark.parametrize('L,H,B,T', [(2, 1024, 1, 16), (2, 1024, 3, 6), (2, 256, 70, 5), (1, 64, 9, 4), (3, 100, 300, 3),
                                      (2, 128, 2100, 2)])
def test_collapsed_forward_equals_the_loop(L, H, B, T, monkeypatch):
    smpl_np = synth.synthetic_smpl(0)
    fast, loop, _ = _pair(monkeypatch, L, H, 41, smpl_np)
    x = torch.from_numpy(synth.synthetic_windows(B, T, 42)).cuda()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        a = fast(x, J_regressor=J)[0]
        b = loop(x, J_regressor=J)[0]
        a49 = fast(x)[0]
        b49 = loop(x)[0]
    for k in KEYS:
        assert (a[k] - b[k]).abs().max().item() < 5e-6, k          # one fp32 rounding of the collapsed matrix vs ten products
        assert (a49[k] - b49[k]).abs().max().item() < 5e-6, k


def test_standalone_regressor_and_the_cases_that_keep_the_loop(monkeypatch):
    """Regressor.forward on given features: n_iter = 3 with the model's initial state is the collapsed product; another n_iter
    or a caller-given initial state runs the loop (and must still match the reference: goldens in test_gpu_geometry.py)."""
    smpl_np = synth.synthetic_smpl(0)
    fast, loop, _ = _pair(monkeypatch, 1, 64, 43, smpl_np)
    feat = torch.from_numpy(synth.normal('col/feat', (11, 2048), std=0.6)).cuda()
    with torch.no_grad():
        for kw in ({}, {'n_iter': 2}, {'n_iter': 0}, {'n_iter': 5},
                   {'init_cam': torch.tensor([[0.8, 0.1, -0.2]]).repeat(11, 1).cuda()}):
            a = fast.regressor(feat, **kw)[0]
            b = loop.regressor(feat, **kw)[0]
            tol = 5e-6 if not kw else 0.0                  # everything but the default call is the same code in both handles
            for k in KEYS:
                assert (a[k] - b[k]).abs().max().item() <= tol, (kw.keys(), k)


def test_collapsed_matrices_follow_the_weights(monkeypatch):
    """In-place weight updates repack, and the repack re-derives the collapsed map (both the regressor's and the tail's)."""
    smpl_np = synth.synthetic_smpl(0)
    fast, loop, _ = _pair(monkeypatch, 2, 128, 44, smpl_np)
    x = torch.from_numpy(synth.synthetic_windows(4, 5, 45)).cuda()
    with torch.no_grad():
        first = fast(x)[0]['theta'].clone()
        for mdl in (fast, loop):
            mdl.regressor.fc2.weight.mul_(1.5)
            mdl.regressor.decpose.bias.add_(0.01)
            mdl.encoder.linear_rec.weight.mul_(0.5)
        a = fast(x)[0]
        b = loop(x)[0]
    assert (a['theta'] - first).abs().max().item() > 1e-3
    for k in KEYS:
        assert (a[k] - b[k]).abs().max().item() < 5e-6, k


def test_exact_fp32_mode_keeps_the_reference_op_order(monkeypatch):
    """TEPOSE_EXACT_FP32=1 handles run the loop on the exact-fp32 MFMA whatever the collapse knob says."""
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    monkeypatch.setenv('TEPOSE_EXACT_FP32', '1')
    a, _, _ = build_model(1, 64, seed=46, device='cuda', smpl_np=smpl_np)
    monkeypatch.setenv('TEPOSE_COLLAPSE_REGRESSOR', '0')
    b, _, _ = build_model(1, 64, seed=46, device='cuda', smpl_np=smpl_np)
    x = torch.from_numpy(synth.synthetic_windows(6, 4, 47)).cuda()
    with torch.no_grad():
        oa, ob = a(x)[0], b(x)[0]
    for k in KEYS:
        assert torch.equal(oa[k], ob[k]), k


@pytest.mark.parametrize('gain', [1.0, 3.0])
def test_collapse_with_trained_like_decoder_norms(gain, monkeypatch):
    """Conditioning: the other cases use decoders at 0.35 / sqrt(fan_in), where G = I + Wd W2 W1b is close
    to I.  Here decpose / decshape / deccam carry xavier-uniform weights of gain 1 and 3 (bound gain * sqrt(6 / (fan_in +
    fan_out)); the reference initialises with gain 0.01, spin.py:222-224, training grows them), ||G||_2 = 1.17 / 1.54,
    ||G^3||_2 = 1.6 / 3.5, and the model's mean pose is far from the identity rotation (N(0,1) 6D rows).  Collapsed map, FC
    loop and the fp64 oracle's loop must agree within 1e-4 on the regressed state (theta carries cam and shape; rotmat the
    pose through rot6d) -- measured: the collapsed product is the closer of the two fp32 paths."""
    import math
    from oracle import tepose_ref as O
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    state = synth.synthetic_state_dict(2, 128, 51)
    for name, rows in (('decpose', 144), ('decshape', 10), ('deccam', 3)):
        bound = gain * math.sqrt(6.0 / (1024 + rows))
        state['regressor.%s.weight' % name] = synth.uniform('cond/%s/%g' % (name, gain), (rows, 1024), -bound, bound)
    state['regressor.init_pose'] = synth.normal('cond/pose', (1, 144), std=1.0)
    state['regressor.init_shape'] = synth.normal('cond/shape', (1, 10), std=1.5)
    fast, _, _ = build_model(2, 128, seed=51, device='cuda', smpl_np=smpl_np, state=state)
    monkeypatch.setenv('TEPOSE_COLLAPSE_REGRESSOR', '0')
    loop, _, _ = build_model(2, 128, seed=51, device='cuda', smpl_np=smpl_np, state=state)
    monkeypatch.delenv('TEPOSE_COLLAPSE_REGRESSOR')
    x = synth.synthetic_windows(9, 5, 52)
    xd = torch.from_numpy(x).cuda()
    with torch.no_grad():
        a, b = fast(xd)[0], loop(xd)[0]
    ref = O.tepose_fwd(state, smpl_np, x, 2, dtype=torch.float64)
    for k in ('theta', 'rotmat', 'kp_3d'):
        r = torch.as_tensor(ref[k]).double().reshape(a[k].shape)
        if k == 'theta':                       # axis-angle is ill-conditioned at pi: compare cam and shape here, the pose via rotmat
            sel = [0, 1, 2] + list(range(75, 85))
            da, db = (a[k].cpu().double() - r)[:, sel].abs().max().item(), (b[k].cpu().double() - r)[:, sel].abs().max().item()
        else:
            da, db = (a[k].cpu().double() - r).abs().max().item(), (b[k].cpu().double() - r).abs().max().item()
        assert da < 1e-4 and db < 1e-4, (k, da, db)
    feat = torch.from_numpy(synth.normal('cond/feat', (70, 2048), std=0.6)).cuda()      # standalone regressor: feat Mf^T + k0
    with torch.no_grad():
        ra, rb = fast.regressor(feat)[0], loop.regressor(feat)[0]
    assert (ra['rotmat'] - rb['rotmat']).abs().max().item() < 1e-4
    assert (ra['theta'][:, 75:] - rb['theta'][:, 75:]).abs().max().item() < 1e-4
