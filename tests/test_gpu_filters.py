"""GPU: post-filter kernels against vectors from the reference's OneEuroFilter / quaternion utilities."""
import os

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_one_euro_and_slerp_match_reference():
    from tepose_amd.filters import one_euro, smooth_pose_mat
    g = np.load(os.path.join(GOLDEN, 'filters.npz'))
    hat = one_euro(g['pose']).cpu().numpy()
    assert hat.shape == g['pose_hat'].shape
    assert np.abs(hat - g['pose_hat']).max() < 1e-5
    Rs = smooth_pose_mat(g['R'], ratio=0.3)
    assert isinstance(Rs, np.ndarray) and np.abs(Rs - g['R_smooth']).max() < 1e-5
    one = smooth_pose_mat(torch.from_numpy(g['R'][:1]).cuda())
    assert (one.cpu() - torch.from_numpy(g['R'][:1])).abs().max() < 1e-6     # single frame: identity filter


def test_smooth_pose_pipeline_vs_oracle():
    from oracle import tepose_ref as O
    from tepose_amd.filters import smooth_pose
    from tepose_amd.smpl import SMPL
    smpl_np = synth.synthetic_smpl(0)
    g = np.load(os.path.join(GOLDEN, 'filters.npz'))
    betas = synth.normal('flt/betas', (50, 10), std=0.5)
    verts, pose_hat, joints = smooth_pose(g['pose'], betas, SMPL.from_tables(smpl_np))
    assert verts.shape == (50, 6890, 3) and joints.shape == (50, 49, 3)
    ph = O.one_euro_filter(g['pose'])
    s = O.smpl_tensors(smpl_np)
    R = O.batch_rodrigues(torch.from_numpy(ph.reshape(-1, 3))).view(50, 24, 3, 3)
    v_ref, posed = O.lbs(s, torch.from_numpy(betas), R)
    assert np.abs(verts - v_ref.numpy()).max() < 1e-4
    assert np.abs(joints - O.smpl_joints49(s, v_ref, posed).numpy()).max() < 1e-4
    # ... and against what the reference's own smooth_pose returned for this very input (lib/utils/smooth_pose.py:24-68, called by the fixture generator)
    assert np.abs(pose_hat.reshape(50, 24, 3) - g['pose_hat']).max() < 1e-5
    assert np.abs(verts[:, ::53] - g['smooth_verts_sub']).max() < 1e-4
    assert np.abs(joints - g['smooth_joints']).max() < 1e-4
