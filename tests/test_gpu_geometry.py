"""GPU: every R -> axis-angle branch and the degenerate rot6d inputs of the reference-generated geometry goldens
(tests/golden/geometry.npz = lib/utils/geometry.py:68-233 and :330-344 run on edge vectors: each quaternion branch,
angle 0, angles at and near pi about six axes, 200 random rotations; zero / parallel 6D inputs), driven through the HIP
code two ways: the standalone geometry entry points, and the regressor kernel itself (Regressor.forward with per-call
init_pose and n_iter=0, lib/models/spin.py:240-251, which runs rot6d -> R -> aa inside smpl_prep_kernel)."""
import os

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'geometry.npz'))


def _rodrigues(aa):
    """axis-angle [N,3] -> rotation matrices, fp64 (well-conditioned everywhere, also at pi)."""
    aa = np.asarray(aa, dtype=np.float64)
    th = np.linalg.norm(aa, axis=1)
    k = aa / np.maximum(th, 1e-300)[:, None]
    K = np.zeros((len(aa), 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    return np.eye(3) + np.sin(th)[:, None, None] * K + (1 - np.cos(th))[:, None, None] * (K @ K)


def _branch(R):
    """which quaternion candidate geometry.py:191-216 selects (on M = R^T)"""
    M = np.transpose(R, (0, 2, 1))
    d2, d01, d0n1 = M[:, 2, 2] < 1e-6, M[:, 0, 0] > M[:, 1, 1], M[:, 0, 0] < -M[:, 1, 1]
    return np.where(d2 & d01, 0, np.where(d2, 1, np.where(d0n1, 2, 3)))


def test_golden_covers_every_branch_and_the_singular_angles():
    b = _branch(G['R'])
    assert set(b.tolist()) == {0, 1, 2, 3}
    ang = np.linalg.norm(G['aa'], axis=1)
    assert (ang == 0).any() and (np.abs(ang - np.pi) < 2e-3).sum() >= 10 and (ang > 3.0).sum() >= 20


def test_rotmat_to_angle_axis_all_golden_rows_direct():
    """Same fp32 op sequence as the reference (sqrt and divide correctly rounded, atan2f within a few ulp):
    every row, including exact pi and pi - 1e-3, compares DIRECTLY -- no angle mask."""
    from tepose_amd.geometry import rotation_matrix_to_angle_axis
    aa = rotation_matrix_to_angle_axis(torch.from_numpy(G['R']).cuda()).cpu().numpy()
    assert np.isfinite(aa).all()
    assert np.abs(aa - G['aa']).max() < 2e-6, np.abs(aa - G['aa']).max()
    for br in range(4):
        sel = _branch(G['R']) == br
        assert np.abs(aa[sel] - G['aa'][sel]).max() < 2e-6, br


def test_rotmat_to_angle_axis_nan_rows_become_zero():
    from tepose_amd.geometry import rotation_matrix_to_angle_axis
    R = torch.eye(3).repeat(3, 1, 1)
    R[1] = float('nan')
    R[2] = 0.0                                      # the all-zero "rotation" of a degenerate rot6d input: branch c1, y = .5
    aa = rotation_matrix_to_angle_axis(R.cuda()).cpu()
    assert torch.equal(aa[:2], torch.zeros(2, 3))   # geometry.py:115 pose[torch.isnan(pose)] = 0.0
    from oracle import tepose_ref as O
    assert (aa[2] - O.rotmat_to_angle_axis(R[2:3])[0]).abs().max() < 1e-6 and abs(float(aa[2, 1]) - np.pi) < 1e-6


def test_rot6d_to_rotmat_golden_incl_degenerate_rows():
    from tepose_amd.geometry import rot6d_to_rotmat
    R6 = rot6d_to_rotmat(torch.from_numpy(G['x6']).cuda()).cpu().numpy()
    assert R6.shape == G['R6'].shape
    assert np.abs(R6 - G['R6']).max() < 1e-6
    # x6[0,:6] = 0 -> b1 = b2 = b3 = 0; x6[1,:6]: a2 parallel to a1 -> b2 = 0 (normalize clamps at eps, no NaN)
    assert np.array_equal(R6[0], np.zeros((3, 3))) and np.isfinite(R6).all()
    assert np.abs(R6[24, :, 1]).max() < 1e-6 and np.abs(R6[24] - G['R6'][24]).max() < 1e-7


def _regressor(smpl_np):
    from tepose_amd.testing import build_model
    model, _, _ = build_model(1, 64, seed=3, device='cuda', smpl_np=smpl_np)
    return model.regressor


def test_regressor_kernel_on_golden_rotations_via_init_pose():
    """All 249 golden rotations through smpl_prep_kernel: init_pose = the 6D form (first two columns) of each R, 24 per
    person, n_iter = 0.  The kernel's rot6d step re-normalises the (unit) columns, which moves R by <= 1 ulp -- so aa is
    compared directly wherever R -> aa is well-conditioned for that perturbation (|angle - pi| > 1e-2) and AS A ROTATION
    for the rows at / next to pi, where a 1-ulp change of R legitimately flips the axis sign."""
    smpl_np = synth.synthetic_smpl(0)
    reg = _regressor(smpl_np)
    R = G['R']
    n = (len(R) + 23) // 24
    Rp = np.concatenate([R, np.tile(np.eye(3, dtype=np.float32), (n * 24 - len(R), 1, 1))]).reshape(n, 24, 3, 3)
    x6 = Rp[:, :, :, :2].reshape(n, 144)            # x.view(-1,3,2): element (r, c) of the first two columns
    ref_aa = np.concatenate([G['aa'], np.zeros((n * 24 - len(R), 3), np.float32)])
    feat = torch.zeros(n, 2048, device='cuda')
    with torch.no_grad():
        out = reg(feat, init_pose=torch.from_numpy(x6).cuda(), n_iter=0)[0]
    rot = out['rotmat'].cpu().numpy().reshape(-1, 3, 3)
    assert np.abs(rot - Rp.reshape(-1, 3, 3)).max() < 5e-7            # rot6d(R[:, :2]) == R
    aa = out['theta'][:, 3:75].cpu().numpy().reshape(-1, 3)
    ang = np.linalg.norm(ref_aa, axis=1)
    well = np.abs(ang - np.pi) > 1e-2
    assert well.sum() > 200 and (~well).sum() >= 10
    assert np.abs(aa[well] - ref_aa[well]).max() < 2e-5
    assert np.abs(_rodrigues(aa) - _rodrigues(ref_aa)).max() < 2e-6   # every row, pi included
    assert np.abs(np.linalg.norm(aa[~well], axis=1) - ang[~well]).max() < 1e-3   # the angle itself is stable at pi
    # init_shape / init_cam pass through untouched
    assert np.abs(out['theta'][:, :3].cpu().numpy() - reg.init_cam.cpu().numpy()).max() == 0


def test_regressor_kernel_on_degenerate_rot6d_rows_and_per_call_init():
    """geometry.npz x6 rows (incl. zero and parallel 6D vectors) as init_pose; plus init_shape / init_cam per call and
    n_iter = 2 against the oracle's regressor loop started from the same state."""
    from oracle import tepose_ref as O
    smpl_np = synth.synthetic_smpl(0)
    reg = _regressor(smpl_np)
    x6 = torch.from_numpy(G['x6'][:8]).cuda()
    with torch.no_grad():
        out = reg(torch.zeros(8, 2048, device='cuda'), init_pose=x6, n_iter=0)[0]
    assert np.abs(out['rotmat'].cpu().numpy().reshape(-1, 3, 3) - G['R6'][:8 * 24]).max() < 1e-6
    assert torch.isfinite(out['theta']).all() and torch.isfinite(out['verts']).all()
    ref_aa = O.rotmat_to_angle_axis(torch.from_numpy(G['R6'][:8 * 24])).reshape(8, 72)
    aa = out['theta'][:, 3:75].cpu()
    assert (aa[0, :6] - ref_aa[0, :6]).abs().max() < 1e-6            # R = 0 (joint 0) and b2 = 0 (joint 1 of row 1)
    assert np.abs(_rodrigues(aa.reshape(-1, 3).numpy()) - _rodrigues(ref_aa.reshape(-1, 3).numpy())).max() < 1e-5

    state = synth.synthetic_state_dict(1, 64, 3)
    _, regw = O.split_state_dict(state, torch.float64)
    smpl = O.smpl_tensors(smpl_np, torch.float64)
    N = 5
    feat = synth.normal('gfeat', (N, 2048), std=0.5)
    ip = synth.normal('gip', (N, 144), std=0.7)
    ish = synth.normal('gis', (N, 10), std=0.5)
    ic = synth.normal('gic', (N, 3), std=0.1) + np.array([0.9, 0, 0], np.float32)
    with torch.no_grad():
        got = reg(torch.from_numpy(feat).cuda(), init_pose=torch.from_numpy(ip).cuda(),
                  init_shape=torch.from_numpy(ish).cuda(), init_cam=torch.from_numpy(ic).cuda(), n_iter=2)[0]
        ref = O.regressor_fwd(regw, smpl, torch.from_numpy(feat).double(), None, n_iter=2,
                              init=(torch.from_numpy(ip).double(), torch.from_numpy(ish).double(),
                                    torch.from_numpy(ic).double()))
    for k in ('rotmat', 'verts', 'kp_3d'):
        assert (got[k].cpu().double() - ref[k]).abs().max() < 1e-4, k
    assert (got['theta'][:, :3].cpu().double() - ref['theta'][:, :3]).abs().max() < 1e-4


@pytest.mark.parametrize('name', ['regressor_init_N5_it2_j14', 'regressor_init_N3_it0_j49'])
def test_regressor_per_call_init_vs_reference_golden(name):
    """The HIP regressor with per-call initial states against vectors produced by the REFERENCE's Regressor class called
    that way (tests/golden/make_golden.py::regressor_init_case)."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', name + '.npz'))
    N, n_iter, use_j, seed_w = [int(v) for v in g['meta']]
    smpl_np = synth.synthetic_smpl(0)
    from tepose_amd.testing import build_model
    model, _, _ = build_model(1, 64, seed=seed_w, device='cuda', smpl_np=smpl_np)
    reg = model.regressor
    feat = torch.from_numpy(synth.normal('gold/feat%d' % N, (N, 2048), std=0.5)).cuda()
    ip = torch.from_numpy(synth.normal('gold/ip%d' % N, (N, 144), std=0.7)).cuda()
    ish = torch.from_numpy(synth.normal('gold/is%d' % N, (N, 10), std=0.5)).cuda()
    ic = (torch.from_numpy(synth.normal('gold/ic%d' % N, (N, 3), std=0.1)) + torch.tensor([0.9, 0., 0.])).cuda()
    J = torch.from_numpy(smpl_np['J_regressor_h36m']) if use_j else None
    with torch.no_grad():
        out = reg(feat, init_pose=ip, init_shape=ish, init_cam=ic, n_iter=n_iter, J_regressor=J)[0]
        only = reg(feat, init_pose=ip, n_iter=n_iter, J_regressor=J)[0]
    o = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.abs(o['rotmat'] - g['rotmat']).max() < 1e-4
    assert np.abs(o['kp_3d'] - g['kp_3d']).max() < 1e-4
    assert np.abs(o['kp_2d'] - g['kp_2d']).max() < 5e-4
    assert np.abs(o['verts'][:, ::53] - g['verts_sub']).max() < 1e-4
    assert np.abs(o['theta'][:, :3] - g['theta'][:, :3]).max() < 1e-4 and np.abs(o['theta'][:, 75:] - g['theta'][:, 75:]).max() < 1e-4
    assert np.abs(_rodrigues(o['theta'][:, 3:75].reshape(-1, 3)) - _rodrigues(g['theta'][:, 3:75].reshape(-1, 3))).max() < 1e-4
    assert np.abs(only['kp_3d'].cpu().numpy() - g['kp_3d_only_pose']).max() < 1e-4
