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


def vibe_golden_config(g):
    meta = [int(v) for v in g['meta']]          # the first two fixtures predate the flags (evaluate.py:93-101's configuration)
    return tuple(meta[:6]) + (tuple(bool(v) for v in meta[6:9]) if len(meta) > 6 else (False, True, True))


def _build(L, H, seed, smpl_np, bidirectional=False, add_linear=True, use_residual=True):
    from tepose_amd.smpl import SMPL
    from tepose_amd.vibe import VIBE
    state = synth.synthetic_vibe_state_dict(L, H, seed, bidirectional=bidirectional, add_linear=add_linear)
    mean = {'pose': state['regressor.init_pose'][0], 'shape': state['regressor.init_shape'][0],
            'cam': state['regressor.init_cam'][0]}
    model = VIBE(seqlen=16, n_layers=L, hidden_size=H, add_linear=add_linear, bidirectional=bidirectional,
                 use_residual=use_residual, pretrained='', smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean)
    sd = model.state_dict()
    assert {k for k in sd if k.startswith('encoder.')} == {k for k in state if k.startswith('encoder.')}   # vibe.py:36-47's keys
    for k, v in state.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    return model.cuda().eval(), state


@pytest.mark.parametrize('name', ['vibe_L2H128_B2N20', 'vibe_L1H64_B1N5', 'vibe_bi_L2H64_B2N7', 'vibe_bi_L1H100_B3N4_nores',
                                  'vibe_nolin_L1H2048_B1N4', 'vibe_nolin_L2H96_B2N6'])
def test_vibe_matches_reference_golden(name, smpl_np):
    """Every constructor configuration of lib/models/vibe.py:27-47 against vectors from the reference's own class:
    uni-directional + linear (what evaluate.py / demo.py build), bidirectional (linear from 2*hidden), no linear with
    hidden = 2048 (residual on the GRU output), no linear with hidden != 2048 (encoder only: no residual, vibe.py:59),
    use_residual = False."""
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    L, H, B, N, seed_w, seed_x, bidir, lin, res = vibe_golden_config(g)
    model, _ = _build(L, H, seed_w, smpl_np, bidir, lin, res)
    x = torch.from_numpy(synth.synthetic_windows(B, N, seed_x)[:, :, :2048].copy()).cuda()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        feat = model.encoder(x)
    assert feat.shape == g['feature'].shape
    assert np.abs(feat.cpu().numpy() - g['feature']).max() < 2e-5
    if 'theta' not in g.files:
        return
    with torch.no_grad():
        out = model(x, J_regressor=J)[-1]
    assert feat.shape == (B, N, 2048) and out['theta'].shape == (B, N, 85)
    assert out['verts'].shape == (B, N, 6890, 3) and out['kp_3d'].shape == (B, N, 14, 3)
    assert np.abs(out['kp_3d'].cpu().numpy() - g['kp_3d']).max() < 1e-4
    assert np.abs(out['rotmat'].cpu().numpy() - g['rotmat']).max() < 1e-4
    assert np.abs(out['verts'].cpu().numpy()[:, :, ::53] - g['verts_sub']).max() < 1e-4


@pytest.mark.parametrize('L,H,B,N,bidir,lin,res', [(2, 200, 5, 9, True, True, True), (3, 64, 2, 3, True, False, False),
                                                    (2, 2048, 2, 3, False, False, True), (1, 130, 4, 1, False, False, True),
                                                    (2, 1024, 3, 12, True, True, True)])
def test_vibe_encoder_configurations_vs_oracle(L, H, B, N, bidir, lin, res, smpl_np):
    from oracle import tepose_ref as O
    model, state = _build(L, H, 21, smpl_np, bidir, lin, res)
    x = synth.synthetic_windows(B, N, 79)[:, :, :2048].copy()
    with torch.no_grad():
        feat = model.encoder(torch.from_numpy(x).cuda())
    enc, _ = O.split_state_dict(state, torch.float64)
    ref = O.vibe_encoder_fwd(enc, torch.from_numpy(x).double(), L, res)
    assert feat.shape == ref.shape == (B, N, 2048 if (lin or bidir) else H)
    assert (feat.cpu().double() - ref).abs().max() < 2e-5


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


def test_vibe_pack_and_blob_follow_the_constructor_flags():
    """n_w of tepose_pack_vibe_encoder is 4*L*D (+2 with the linear); a blob packed by one configuration is refused by a
    handle of another (the header records the flags)."""
    from tepose_amd import _lib
    from tepose_amd.engine import Engine
    from tepose_amd.vibe import TemporalEncoder
    lib = _lib.load()
    enc = TemporalEncoder(n_layers=1, hidden_size=64, bidirectional=True, add_linear=False).cuda()
    assert enc.linear is not None and enc.linear.in_features == 128           # vibe.py:44-45
    x = torch.randn(1, 3, 2048, device='cuda')
    with torch.no_grad():
        assert enc(x).shape == (1, 3, 2048)
    assert lib.tepose_vibe_feature_dim(enc._engine.handle) == 2048
    plain = TemporalEncoder(n_layers=1, hidden_size=64, add_linear=False).cuda()
    assert plain.linear is None and lib.tepose_vibe_feature_dim(plain._engine.handle) == 64
    with torch.no_grad():
        assert plain(x).shape == (1, 3, 64)
    blob = enc._engine.blob
    for other in (Engine(1, 64, 'vibe'), Engine(1, 64, 'vibe', add_linear=False)):
        big = torch.zeros(max(other.packed_bytes, blob.numel()), dtype=torch.uint8, device='cuda')
        big[:blob.numel()] = blob
        assert lib.tepose_set_blob(other.handle, big.data_ptr(), big.numel()) == 0
        assert lib.tepose_adopt_blob(other.handle) == -4
    ws = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
    arr = _lib.ptr_array([ws.data_ptr()] * 6)
    assert lib.tepose_pack_vibe_encoder(enc._engine.handle, arr, 6, None) == -1      # bidirectional, L = 1: 8 + 2 tensors
