"""GPU: VIBE bootstrap model (tepose_amd/vibe.py) against vectors from the reference's
lib.models.vibe.VIBE and against the oracle."""
import os

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def smpl_np():
    return synth.synthetic_smpl(0)


def _build(L, H, seed, smpl_np):
    from tepose_amd.smpl import SMPL
    from tepose_amd.vibe import VIBE
    state = synth.synthetic_vibe_state_dict(L, H, seed)
    mean = {'pose': state['regressor.init_pose'][0], 'shape': state['regressor.init_shape'][0],
            'cam': state['regressor.init_cam'][0]}
    model = VIBE(seqlen=16, n_layers=L, hidden_size=H, add_linear=True, bidirectional=False, use_residual=True,
                 pretrained='', smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean)
    sd = model.state_dict()
    for k, v in state.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    return model.cuda().eval(), state


@pytest.mark.parametrize('name', ['vibe_L2H128_B2N20', 'vibe_L1H64_B1N5'])
def test_vibe_matches_reference_golden(name, smpl_np):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    L, H, B, N, seed_w, seed_x = [int(v) for v in g['meta']]
    model, _ = _build(L, H, seed_w, smpl_np)
    x = torch.from_numpy(synth.synthetic_windows(B, N, seed_x)[:, :, :2048].copy()).cuda()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        feat = model.encoder(x)
        out = model(x, J_regressor=J)[-1]
    assert feat.shape == (B, N, 2048) and out['theta'].shape == (B, N, 85)
    assert out['verts'].shape == (B, N, 6890, 3) and out['kp_3d'].shape == (B, N, 14, 3)
    assert np.abs(feat.cpu().numpy() - g['feature']).max() < 2e-5
    assert np.abs(out['kp_3d'].cpu().numpy() - g['kp_3d']).max() < 1e-4
    assert np.abs(out['rotmat'].cpu().numpy() - g['rotmat']).max() < 1e-4
    assert np.abs(out['verts'].cpu().numpy()[:, :, ::53] - g['verts_sub']).max() < 1e-4


def test_vibe_published_size_vs_oracle(smpl_np):
    """n_layers=2, hidden=1024 (evaluate.py:93-101) on a 40-frame tracklet."""
    from oracle import tepose_ref as O
    model, state = _build(2, 1024, 3, smpl_np)
    x = synth.synthetic_windows(1, 40, 77)[:, :, :2048].copy()
    with torch.no_grad():
        out = model(torch.from_numpy(x).cuda())[-1]
    ref = O.vibe_fwd(state, smpl_np, x, 2)
    assert out['kp_3d'].shape == (1, 40, 49, 3)
    assert (out['verts'].cpu().reshape(-1, 6890, 3) - ref['verts']).abs().max() < 1e-4
    assert (out['kp_3d'].cpu().reshape(-1, 49, 3) - ref['kp_3d']).abs().max() < 1e-4


def test_vibe_long_tracklets_split_regressor(smpl_np):
    """3 tracklets x 50 frames = 150 regressor rows (> 96): the FC stack and the blend-shape GEMM run on the
    split-precision kernel; same tolerance against the fp64 oracle."""
    from oracle import tepose_ref as O
    model, state = _build(1, 64, 5, smpl_np)
    x = synth.synthetic_windows(3, 50, 78)[:, :, :2048].copy()
    with torch.no_grad():
        out = model(torch.from_numpy(x).cuda())[-1]
    ref = O.vibe_fwd(state, smpl_np, x, 1)
    assert (out['verts'].cpu().reshape(-1, 6890, 3) - ref['verts']).abs().max() < 1e-4
    assert (out['kp_3d'].cpu().reshape(-1, 49, 3) - ref['kp_3d']).abs().max() < 1e-4
    assert (out['theta'].cpu().reshape(-1, 85)[:, :3] - ref['theta'][:, :3]).abs().max() < 1e-4


def test_vibe_rejects_unsupported_configs():
    from tepose_amd.vibe import TemporalEncoder
    with pytest.raises(NotImplementedError):
        TemporalEncoder(n_layers=1, hidden_size=64, bidirectional=True, add_linear=True)
