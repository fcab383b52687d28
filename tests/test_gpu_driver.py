"""GPU: the autoregressive window driver batched over clips (tepose_amd/driver.py) against the
reference's per-clip loop (golden vectors from evaluate.py:247-269 semantics) and the oracle."""
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


def _aa_close(a, b, tol):
    a, b = a.reshape(-1, 3), b.reshape(-1, 3)
    ok = np.linalg.norm(b, axis=1) < 3.0
    return np.abs(a[ok] - b[ok]).max() < tol and ok.mean() > 0.8


@pytest.mark.parametrize('name', ['driver_L2H128_N40T6', 'driver_L1H64_N9T4'])
def test_single_clip_matches_reference_loop(name, smpl_np):
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    L, H, N, T, seed_w, seed_x = [int(v) for v in g['meta']]
    model, _, _ = build_model(L, H, seed=seed_w, device='cuda', smpl_np=smpl_np)
    w = synth.synthetic_windows(1, N, seed_x)[0]
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    out = run_clips(model, [torch.from_numpy(w[:, :2048].copy())], [torch.from_numpy(g['theta_init'])], T,
                    J_regressor=J)[0]
    assert out['theta'].shape == (N - T + 1, 85)
    # 35 feedback steps: errors of window j feed window j+1; budget stays 1e-4
    assert np.abs(out['kp_3d'].cpu().numpy() - g['kp_3d']).max() < 1e-4
    assert np.abs(out['verts'].cpu().numpy()[:, ::53] - g['verts_sub']).max() < 1e-4
    th = out['theta'].cpu().numpy()
    assert np.abs(th[:, :3] - g['theta'][:, :3]).max() < 1e-4
    assert np.abs(th[:, 75:] - g['theta'][:, 75:]).max() < 1e-4
    assert _aa_close(th[:, 3:75], g['theta'][:, 3:75], 1e-4)


def test_ragged_clips_in_lockstep_equal_per_clip_runs(smpl_np):
    """Clips of different lengths (one shorter than the window) advance together; each must equal
    its own B=1 oracle run, and the result order must follow the input order."""
    from oracle import tepose_ref as O
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    L, H, T = 1, 64, 5
    model, state, _ = build_model(L, H, seed=9, device='cuda', smpl_np=smpl_np)
    lens = [9, 3, 14, 5, 11]
    feats, inits = [], []
    for i, n in enumerate(lens):
        w = synth.synthetic_windows(1, max(n, T), 600 + i)[0]
        feats.append(torch.from_numpy(w[:n, :2048].copy()))
        th = w[:T - 1, 2048:].copy()
        th[:, :3] = [1, 0, 0]
        inits.append(torch.from_numpy(th))
    res = run_clips(model, feats, inits, T)
    assert res[1] is None                                   # shorter than the window: skipped
    for i, n in enumerate(lens):
        if n < T:
            continue
        ref = O.run_clip(state, smpl_np, feats[i].numpy(), inits[i].numpy(), T, L)
        assert res[i]['theta'].shape[0] == n - T + 1
        assert (res[i]['verts'].cpu() - ref['verts']).abs().max() < 1e-4
        assert (res[i]['kp_3d'].cpu() - ref['kp_3d']).abs().max() < 1e-4
        assert (res[i]['rotmat'].cpu() - ref['rotmat']).abs().max() < 1e-4


@pytest.mark.parametrize('L,H,T', [(2, 128, 6), (1, 64, 4), (2, 64, 2)])
def test_cached_projections_equal_uncached(L, H, T, smpl_np):
    """The projection cache (ring of layer-0 gate pre-activations) is an exact re-association of the
    same GEMM rows: per-clip results must agree with the uncached driver to rounding."""
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    model, _, _ = build_model(L, H, seed=13, device='cuda', smpl_np=smpl_np)
    lens = [T + 9, T, T + 3, T - 1 if T > 1 else 1, T + 14]
    feats, inits = [], []
    for i, n in enumerate(lens):
        w = synth.synthetic_windows(1, max(n, T) + 1, 800 + i)[0]
        feats.append(torch.from_numpy(w[:n, :2048].copy()))
        inits.append(torch.from_numpy(w[:T - 1, 2048:].copy()))
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    a = run_clips(model, feats, inits, T, J_regressor=J, cache_projections=False)
    b = run_clips(model, feats, inits, T, J_regressor=J, cache_projections=True)
    for ra, rb in zip(a, b):
        assert (ra is None) == (rb is None)
        if ra is None:
            continue
        for k in ra:
            assert ra[k].shape == rb[k].shape
            assert (ra[k] - rb[k]).abs().max() < 2e-5, k


def test_cached_projections_many_clips_split_path(smpl_np):
    """> 4 concurrent clips (here 801): window batches, the per-frame projection of the cache and the regressor all run on the
    split-precision kernels; cached and uncached drivers must still agree to rounding."""
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    T, C = 4, 801
    model, _, _ = build_model(2, 64, seed=21, device='cuda', smpl_np=smpl_np)
    w = synth.synthetic_windows(C, T + 3, 77)
    feats = [torch.from_numpy(w[i, :T + 2 + (i % 2), :2048].copy()) for i in range(C)]
    inits = [torch.from_numpy(w[i, :T - 1, 2048:].copy()) for i in range(C)]
    a = run_clips(model, feats, inits, T, keep=('theta', 'kp_3d'), cache_projections=False)
    b = run_clips(model, feats, inits, T, keep=('theta', 'kp_3d'), cache_projections=True)
    for i in (0, 1, 400, 800):
        for k in ('theta', 'kp_3d'):
            assert a[i][k].shape == b[i][k].shape
            assert (a[i][k] - b[i][k]).abs().max() < 2e-5, (i, k)


def test_live_stream_window_32_single_clip_vs_oracle_loop(smpl_np):
    """BASELINE.json config 5: seqlen 32, one clip, window advancing one frame at a time with theta feedback
    (demo.py:238-252); every step against the oracle's per-clip loop, cached and uncached drivers."""
    from oracle import tepose_ref as O
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    L, H, T, N = 2, 128, 32, 44
    model, state, _ = build_model(L, H, seed=17, device='cuda', smpl_np=smpl_np)
    w = synth.synthetic_windows(1, N, 901)[0]
    feats = [torch.from_numpy(w[:, :2048].copy())]
    th = w[:T - 1, 2048:].copy()
    inits = [torch.from_numpy(th)]
    ref = O.run_clip(state, smpl_np, feats[0].numpy(), th, T, L)
    for cache in (False, True):
        out = run_clips(model, feats, inits, T, cache_projections=cache)[0]
        assert out['theta'].shape == (N - T + 1, 85)
        for k in ('verts', 'kp_3d', 'rotmat'):
            assert (out[k].cpu() - ref[k]).abs().max() < 1e-4, (cache, k)
        assert (out['theta'][:, :3].cpu() - ref['theta'][:, :3]).abs().max() < 1e-4
        assert (out['theta'][:, 75:].cpu() - ref['theta'][:, 75:]).abs().max() < 1e-4


@pytest.mark.parametrize('name', ['padded_L2H128_T5', 'padded_ds_L1H64_T5', 'padded_ds_h36m_L1H64_T4'])
def test_padded_validation_batch_matches_reference_trainer_loop(name):
    """tepose_amd.driver.validate_padded = lib/core/trainer.py:307-357 on one batch of the validation Datasets
    (zero-padded clips, float16-staged arrays, vidlen_each): accumulators in the trainer's order against vectors from the
    reference model run through that very loop (padding windows included)."""
    import os
    from tepose_amd.driver import validate_padded
    from tepose_amd.testing import build_model
    # (padded_ds_*: the batch is what the reference's validation Dataset + a DataLoader emitted from a database file)
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', name + '.npz'))
    L, H, T, seed_w, seed_x = [int(v) for v in g['meta'][:5]]
    lens = [int(v) for v in g['meta'][5:]]
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(L, H, seed=seed_w, device='cuda', smpl_np=smpl_np)
    target = {'features': torch.from_numpy(g['features'].astype(np.float32)).cuda(),
              'theta_pseu': torch.from_numpy(g['theta_pseu'].astype(np.float32)).cuda(),
              'vidlen_each': torch.tensor(lens).float().view(-1, 1), 'index': torch.arange(len(lens)).float().view(-1, 1)}
    out = validate_padded(model, target, T, J_regressor=torch.from_numpy(smpl_np['J_regressor_h36m']))
    assert out['pred_kp_3d'].shape == g['pred_j3d'].shape and out['pred_j3d_tsr'].shape == g['pred_j3d_tsr'].shape
    assert np.abs(out['pred_kp_3d'].cpu().numpy() - g['pred_j3d']).max() < 1e-4
    assert np.abs(out['pred_theta'].cpu().numpy()[:, 75:] - g['pred_theta'][:, 75:]).max() < 1e-4
    assert np.abs(out['pred_verts'].cpu().numpy()[:, ::53] - g['pred_verts_sub']).max() < 1e-4
    tsr = out['pred_j3d_tsr'].cpu().numpy()
    for c, n in enumerate(lens):
        assert np.abs(tsr[c, T - 1:n] - g['pred_j3d_tsr'][c, T - 1:n]).max() < 1e-4
        assert not tsr[c, :T - 1].any() and not tsr[c, n:].any()
    # Trainer.evaluate on these accumulators (trainer.py:437-503) against what the reference's unbound Trainer.evaluate returned
    from tepose_amd.metrics import trainer_evaluate
    target['kp_3d'] = torch.from_numpy(g['kp_3d'].astype(np.float32))
    ev = trainer_evaluate(model, out, target, T)
    ref = g['eval_mpjpe_pa_accel_accelerr']
    for k, r in zip(('mpjpe', 'pa-mpjpe', 'accel', 'accel_err'), ref):
        assert abs(ev[k] - r) < 2e-4 * abs(r) + 2e-2, (k, ev[k], r)           # mm
    target['theta'] = torch.from_numpy(g['theta'].astype(np.float32))         # ground-truth thetas of the batch: PVE as Trainer.evaluate's own value (trainer.py:479-482)
    ev = trainer_evaluate(model, out, target, T)
    assert abs(ev['pve'] - float(g['eval_pve'])) < 2e-4 * float(g['eval_pve']) + 2e-2, (ev['pve'], float(g['eval_pve']))
    # a clip shorter than the window contributes nothing (split_into_videos_val drops it, _img_utils.py:369-370)
    short = [3 if c == 1 else n for c, n in enumerate(lens)]
    target['vidlen_each'] = torch.tensor(short).float().view(-1, 1)
    out2 = validate_padded(model, target, T)
    assert out2['pred_kp_3d'].shape[0] == sum(max(n - T + 1, 0) for n in short)
    assert out2['pred_kp_3d'].shape[1] == 49


@pytest.mark.parametrize('H,B', [(64, 1), (64, 3), (64, 20), (64, 37), (64, 100), (64, 500), (1024, 37)])
def test_frame_pair_projection_is_the_two_projections(H, B, smpl_np):
    """tepose_project_frame_pair (ONE product of 2 B gathered rows: previous newest frame with its theta -> ring slot, newest frame with zero theta -> the
    `newest` rows; evaluate.py:248-252) against two tepose_project_frames calls.  The same GEMM rows on the same operands; what may differ is how many
    waves the width-first kernel splits K over (4 or 8, by rows and width), i.e. the grouping of the partial sums: equal to rounding everywhere, and bit
    for bit where both run the same instantiation (the published width) or fall back to the two calls (B <= 4: exact-fp32 kernels; 2 B > 768)."""
    from tepose_amd.engine import on_device
    from tepose_amd.testing import build_model
    model, _, _ = build_model(2, H, seed=4, device='cuda', smpl_np=smpl_np)
    eng = model._engine
    dev = torch.device('cuda', 0)
    w = torch.from_numpy(synth.synthetic_windows(B, 2, 31)).to(dev)              # [B, 2 frames, 2133]
    F = w[:, :, :2048].contiguous()
    TH = w[:, 0, 2048:].contiguous()
    with on_device(dev):
        eng.pack_encoder(model.encoder, dev)
        gw = eng.gate_width
        ws = torch.empty(int(eng.lib.tepose_project_frames_workspace_bytes(eng.handle, 2 * B)), dtype=torch.uint8, device=dev)
        a_prev, a_new = torch.full((B, gw), float('nan'), device=dev), torch.full((B, 3, gw), float('nan'), device=dev)
        b_prev, b_new = torch.full((B, gw), float('nan'), device=dev), torch.full((B, 3, gw), float('nan'), device=dev)
        eng.project_frames(F[:, 0].data_ptr(), F.stride(0), TH.data_ptr(), TH.stride(0), B, a_prev.data_ptr(), a_prev.stride(0), ws)
        eng.project_frames(F[:, 1].data_ptr(), F.stride(0), None, 0, B, a_new[:, 1].data_ptr(), a_new.stride(0), ws)
        eng.project_frame_pair(F[:, 0].data_ptr(), F[:, 1].data_ptr(), F.stride(0), TH.data_ptr(), TH.stride(0), B, b_prev.data_ptr(), b_prev.stride(0),
                               b_new[:, 1].data_ptr(), b_new.stride(0), ws)
    torch.cuda.synchronize()
    assert torch.isfinite(b_prev).all() and torch.isfinite(b_new[:, 1]).all()
    assert float((a_prev - b_prev).abs().max()) < 5e-6 and float((a_new[:, 1] - b_new[:, 1]).abs().max()) < 5e-6
    if B <= 4 or 2 * B > 768 or H == 1024:
        assert torch.equal(a_prev, b_prev) and torch.equal(a_new[:, 1], b_new[:, 1])
    assert torch.isnan(b_new[:, 0]).all() and torch.isnan(b_new[:, 2]).all()      # the strided destination: nothing beside its rows is touched
