"""GPU: metrics kernels against values from the reference's eval_utils (golden) and the oracle;
tolerance 0.01 mm."""
import os

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('tag,pel', [('lsp14', 'lsp'), ('mpii17', 'mpii3d')])
def test_joint_metrics_match_reference(tag, pel):
    from tepose_amd.metrics import joint_metrics
    g = np.load(os.path.join(GOLDEN, 'metrics.npz'))
    m = joint_metrics(torch.from_numpy(g[tag + '_pred']).cuda(), torch.from_numpy(g[tag + '_target']).cuda(), pel)
    assert np.abs(m['mpjpe'].cpu().numpy() - g[tag + '_mpjpe']).max() < 1e-2
    assert np.abs(m['pa_mpjpe'].cpu().numpy() - g[tag + '_pa']).max() < 1e-2
    assert np.abs(m['accel'].cpu().numpy() - g[tag + '_accel']).max() < 1e-2


def test_procrustes_edge_cases_vs_oracle():
    """Reflections, planar and near-degenerate point sets, identical inputs, single-frame clips."""
    from oracle import tepose_ref as O
    from tepose_amd.metrics import joint_metrics
    t = torch.from_numpy(synth.normal('pe/t', (40, 14, 3), std=0.3)).double()
    p = t.clone() + torch.from_numpy(synth.normal('pe/n', (40, 14, 3), std=0.02)).double()
    p[0] = t[0]                                  # identical -> 0
    p[1] = t[1] * torch.tensor([1., 1., -1.])    # mirrored: best proper rotation, not a reflection
    t[2, :, 2] = 0.0; p[2, :, 2] = 0.0           # planar sets (rank-2 covariance)
    p[3] = t[3] * 3.0 + 1.0                      # pure similarity -> PA error 0
    m = joint_metrics(p.float().cuda(), t.float().cuda())
    ref = O.joint_metrics(p, t)
    assert abs(float(m['mpjpe'][0])) < 1e-4 and abs(float(m['pa_mpjpe'][3])) < 1e-2
    assert np.abs(m['pa_mpjpe'].cpu().numpy() - ref['pa_mpjpe'].numpy()).max() < 1e-2
    assert np.abs(m['mpjpe'].cpu().numpy() - ref['mpjpe'].numpy()).max() < 1e-2
    one = joint_metrics(p[:1].float().cuda(), t[:1].float().cuda())
    assert float(one['accel'][0]) == 0.0


def test_mpvpe_and_clip_records():
    from oracle import tepose_ref as O
    from tepose_amd.metrics import clip_record, gt_vertices, joint_metrics, reduce_records, vertex_metric
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(1, 64, seed=1, device='cuda', smpl_np=smpl_np)
    theta = synth.synthetic_windows(1, 30, 31)[0, :29, 2048:]              # [29,85] plausible thetas
    gt = gt_vertices(model, torch.from_numpy(theta).cuda())
    ref = O.verts_from_theta(smpl_np, theta)
    assert (gt.cpu() - ref).abs().max() < 1e-4
    pred = gt + 0.01
    mp = vertex_metric(pred, gt)
    assert np.abs(mp.cpu().numpy() - 1000 * 0.01 * 3 ** 0.5).max() < 1e-2
    # records -> frame-weighted means as evaluate.py:461
    recs, all_mpjpe, all_acc = [], [], []
    for cid, n in enumerate((12, 5, 29)):
        t = torch.from_numpy(synth.normal('rec/t%d' % cid, (n, 14, 3), std=0.3)).cuda()
        p = t + torch.from_numpy(synth.normal('rec/n%d' % cid, (n, 14, 3), std=0.03)).cuda()
        m = joint_metrics(p, t)
        recs.append(clip_record(cid, m, mpvpe=mp[:n]))
        all_mpjpe.append(m['mpjpe'].cpu().numpy())
        all_acc.append(m['accel'].cpu().numpy()[1:-1])
    out = reduce_records(torch.stack(recs))
    assert abs(out['mpjpe'] - np.mean(np.concatenate(all_mpjpe))) < 1e-3
    assert abs(out['accel_err'] - np.mean(np.concatenate(all_acc))) < 1e-3
    assert 'mpvpe' in out


def test_smpl_module_is_callable_like_the_reference_class():
    """SMPL(betas, body_pose, global_orient, pose2rot) as smooth_pose.py:35-64 (axis-angle, CPU
    tensors) and evaluate.py:279-286 (rotation matrices) call it."""
    from oracle import tepose_ref as O
    from tepose_amd.smpl import SMPL
    smpl_np = synth.synthetic_smpl(0)
    smpl = SMPL.from_tables(smpl_np)                       # stays on the CPU like in smooth_pose.py
    theta = synth.synthetic_windows(1, 8, 77)[0, :7, 2048:]
    pose = torch.from_numpy(theta[:, 3:75].copy()).view(7, 24, 3)
    betas = torch.from_numpy(theta[:, 75:].copy())
    out = smpl(betas=betas, body_pose=pose[:, 1:], global_orient=pose[:, 0:1])
    assert out.vertices.device.type == 'cpu' and out.vertices.shape == (7, 6890, 3) and out.joints.shape == (7, 49, 3)
    s = O.smpl_tensors(smpl_np)
    R = O.batch_rodrigues(pose.reshape(-1, 3)).view(7, 24, 3, 3)
    v_ref, posed = O.lbs(s, betas, R)
    j_ref = O.smpl_joints49(s, v_ref, posed)
    assert (out.vertices - v_ref).abs().max() < 1e-4
    assert (out.joints - j_ref).abs().max() < 1e-4
    out2 = smpl.cuda()(betas=betas.cuda(), body_pose=R[:, 1:].cuda(), global_orient=R[:, 0:1].cuda(), pose2rot=False)
    assert out2.vertices.is_cuda
    assert (out2.vertices.cpu() - v_ref).abs().max() < 1e-4
    assert (out2.joints.cpu() - j_ref).abs().max() < 1e-4


@pytest.mark.parametrize('n', [1, 2, 3, 4, 5])
def test_smpl_few_persons_every_input_mode(n):
    """1-4 persons run prep + blend shapes + skinning as one launch (csrc/smpl.hip smpl_small_kernel), 5 the four-launch
    chain: axis-angle input (pose2rot=True), rotation-matrix input (evaluate.py:279-286) and theta rows (MPVPE ground truth,
    eval_utils.py:155-169) against the oracle, and the two paths against each other on the same persons."""
    from oracle import tepose_ref as O
    from tepose_amd.metrics import gt_vertices
    from tepose_amd.smpl import SMPL
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    smpl = SMPL.from_tables(smpl_np).cuda()
    theta = synth.synthetic_windows(1, 12, 91)[0, :11, 2048:]
    theta[1, 3:75] *= 9.0                                   # a far-from-rest pose among them
    pose = torch.from_numpy(theta[:, 3:75].copy()).view(11, 24, 3)
    betas = torch.from_numpy(theta[:, 75:].copy())
    s = O.smpl_tensors(smpl_np)
    R = O.batch_rodrigues(pose.reshape(-1, 3)).view(11, 24, 3, 3)
    v_ref, posed = O.lbs(s, betas, R)
    j_ref = O.smpl_joints49(s, v_ref, posed)
    a = smpl(betas=betas[:n].cuda(), body_pose=pose[:n, 1:].cuda(), global_orient=pose[:n, 0:1].cuda())
    b = smpl(betas=betas[:n].cuda(), body_pose=R[:n, 1:].cuda(), global_orient=R[:n, 0:1].cuda(), pose2rot=False)
    for out in (a, b):
        assert out.vertices.shape == (n, 6890, 3) and out.joints.shape == (n, 49, 3)
        assert (out.vertices.cpu() - v_ref[:n]).abs().max() < 1e-4
        assert (out.joints.cpu() - j_ref[:n]).abs().max() < 1e-4
    big = smpl(betas=betas.cuda(), body_pose=pose[:, 1:].cuda(), global_orient=pose[:, 0:1].cuda())       # 11 persons: the chain
    assert (big.vertices[:n] - a.vertices).abs().max() < 1e-5
    model, _, _ = build_model(1, 64, seed=1, device='cuda', smpl_np=smpl_np)
    gt = gt_vertices(model, torch.from_numpy(theta[:n].copy()).cuda())
    assert (gt.cpu() - O.verts_from_theta(smpl_np, theta[:n])).abs().max() < 1e-4


def test_axis_angle_input_matches_the_reference_held_rodrigues():
    """pose2rot=True paths (standalone SMPL call, MPVPE ground-truth mesh) against rotations made by the REFERENCE's own
    batch_rodrigues (lib/utils/geometry.py:22-65; tests/golden/geometry.npz `rod_*`: 12 persons x 24 joints with angles 0,
    1e-9, 1e-4, random, pi -+ 1e-3, beyond 2 pi, both signs) pushed through oracle.lbs in float64.  1e-5 on vertices and
    joints (the library's Rodrigues is the smplx closed form: 2e-8 from the reference's in exact arithmetic)."""
    from oracle import tepose_ref as O
    from tepose_amd.metrics import gt_vertices
    from tepose_amd.smpl import SMPL
    from tepose_amd.testing import build_model
    g = np.load(os.path.join(GOLDEN, 'geometry.npz'))
    smpl_np = synth.synthetic_smpl(0)
    pose = torch.from_numpy(g['rod_aa']).view(12, 24, 3)
    betas = torch.from_numpy(synth.normal('rod/betas', (12, 10), std=0.5))
    s = O.smpl_tensors(smpl_np, torch.float64)
    v_ref, posed = O.lbs(s, betas.double(), torch.from_numpy(g['rod_R64']).view(12, 24, 3, 3))
    j_ref = O.smpl_joints49(s, v_ref, posed)
    smpl = SMPL.from_tables(smpl_np).cuda()
    for n in (12, 3):                                       # the four-launch chain and the one-launch small form
        out = smpl(betas=betas[:n].cuda(), body_pose=pose[:n, 1:].cuda(), global_orient=pose[:n, 0:1].cuda())
        assert (out.vertices.cpu().double() - v_ref[:n]).abs().max() < 1e-5
        assert (out.joints.cpu().double() - j_ref[:n]).abs().max() < 1e-5
    model, _, _ = build_model(1, 64, seed=1, device='cuda', smpl_np=smpl_np)
    theta = torch.cat([torch.tensor([[1., 0., 0.]]).expand(12, 3), pose.reshape(12, 72), betas], dim=1).contiguous()
    gt = gt_vertices(model, theta.cuda())                   # tepose_smpl_verts_from_theta (eval_utils.py:155-169)
    assert (gt.cpu().double() - v_ref).abs().max() < 1e-5
