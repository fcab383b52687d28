"""GPU: the frame-at-a-time live-stream session (tepose_amd/stream.py; reference loop demo.py:238-252, BASELINE config 5).
`StreamSession.push(feature)` = one captured hipGraph replay per arriving frame; it must reproduce the clip-at-once driver
(`run_clips`, itself pinned by the reference-loop goldens) bit for bit, stay within 1e-4 of the reference golden through the
theta feedback, and survive a give-up of the persistent kernels inside a replayed graph."""
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


def _stream_clip(model, w, theta_init, T, J, graph=True, keep=('theta', 'kp_3d', 'verts')):
    from tepose_amd.stream import StreamSession
    feats = torch.from_numpy(w[:, :2048].copy())
    ses = StreamSession(model, T, feats[:T - 1], torch.from_numpy(theta_init), J_regressor=J, keep=keep, graph=graph)
    out = {k: [] for k in keep}
    for f in feats[T - 1:]:
        r = ses.push(f)
        for k in keep:
            out[k].append(r[k].clone())
    return {k: torch.stack(v) for k, v in out.items()}, ses


@pytest.mark.parametrize('name', ['driver_L2H128_N40T6', 'driver_L1H64_N9T4'])
def test_stream_session_equals_run_clips_and_the_reference_golden(name, smpl_np):
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    L, H, N, T, seed_w, seed_x = [int(v) for v in g['meta']]
    model, _, _ = build_model(L, H, seed=seed_w, device='cuda', smpl_np=smpl_np)
    w = synth.synthetic_windows(1, N, seed_x)[0]
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    ref = run_clips(model, [torch.from_numpy(w[:, :2048].copy())], [torch.from_numpy(g['theta_init'])], T, J_regressor=J,
                    keep=('theta', 'kp_3d', 'verts'))[0]
    got, ses = _stream_clip(model, w, g['theta_init'], T, J)
    assert ses.graph is not None and ses.frames == N - T + 1
    for k in ('theta', 'kp_3d', 'verts'):
        assert torch.equal(got[k], ref[k].cpu()), k                       # same forward on the same window, frame after frame
    assert np.abs(got['kp_3d'].numpy() - g['kp_3d']).max() < 1e-4         # 35 feedback steps against the reference's own loop
    assert np.abs(got['verts'].numpy()[:, ::53] - g['verts_sub']).max() < 1e-4
    assert np.abs(got['theta'].numpy()[:, :3] - g['theta'][:, :3]).max() < 1e-4
    assert np.abs(got['theta'].numpy()[:, 75:] - g['theta'][:, 75:]).max() < 1e-4
    # the same session without a graph (eager step) gives the same bits
    eager, ses2 = _stream_clip(model, w, g['theta_init'], T, J, graph=False)
    assert ses2.graph is None
    for k in got:
        assert torch.equal(eager[k], got[k]), k


def test_stream_session_at_the_live_stream_configuration(smpl_np):
    """BASELINE config 5: seqlen 32, batch 1, the published architecture (n_layers 2, hidden 1024), no joint regressor
    (demo.py passes none: 49 joints)."""
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    T, N = 32, 56
    model, _, _ = build_model(2, 1024, seed=3, device='cuda', smpl_np=smpl_np, seqlen=T)
    w = synth.synthetic_windows(1, N, 77)[0]
    th0 = w[:T - 1, 2048:].copy()
    ref = run_clips(model, [torch.from_numpy(w[:, :2048].copy())], [torch.from_numpy(th0)], T, keep=('theta', 'kp_3d'))[0]
    got, ses = _stream_clip(model, w, th0, T, None, keep=('theta', 'kp_3d'))
    assert got['kp_3d'].shape == (N - T + 1, 49, 3)
    for k in got:
        assert torch.isfinite(got[k]).all()
        assert torch.equal(got[k], ref[k].cpu()), k


def test_a_give_up_inside_the_replayed_graph_is_repaired(smpl_np, monkeypatch):
    """The persistent recurrent launches of this handle start giving up AFTER the session's graph has been captured (the test
    hook is set between warm-up and capture), so the first replay returns NaN.  push notices (fault word after its wait),
    switches the model to the step kernels, re-captures, recomputes the frame from the saved window: the stream's results equal
    a healthy model's run to rounding and are never NaN."""
    from tepose_amd.driver import run_clips
    from tepose_amd.stream import StreamSession
    from tepose_amd.testing import build_model
    L, H, T, N = 2, 256, 6, 14
    good, _, _ = build_model(L, H, seed=21, device='cuda', smpl_np=smpl_np)
    monkeypatch.setenv('TEPOSE_SEQ_SPIN_LIMIT', '20000')         # ~20 ms instead of ~2 s per give-up
    bad, _, _ = build_model(L, H, seed=21, device='cuda', smpl_np=smpl_np)
    monkeypatch.delenv('TEPOSE_SEQ_SPIN_LIMIT')
    w = synth.synthetic_windows(1, N, 78)[0]
    th0 = w[:T - 1, 2048:].copy()
    feats = torch.from_numpy(w[:, :2048].copy())
    ref = run_clips(good, [feats], [torch.from_numpy(th0)], T, keep=('theta', 'verts'))[0]
    ses = StreamSession(bad, T, feats[:T - 1], torch.from_numpy(th0), keep=('theta', 'verts'), graph=False)   # healthy warm-up
    eng = bad._engine
    assert eng.uses_persistent(1) and not eng.degraded
    assert eng.lib.tepose_debug_set_test_fault(eng.handle, 1) == 0
    ses.use_graph = True
    with torch.no_grad(), torch.cuda.stream(ses.stream):
        ses._capture()                                            # kernel arguments (the unmeetable wait) are baked in here
    ses.stream.synchronize()
    assert eng.lib.tepose_debug_set_test_fault(eng.handle, 0) == 0   # (the re-captured graph uses the step kernels anyway)
    for j, f in enumerate(feats[T - 1:]):
        if j == 0:
            with pytest.warns(RuntimeWarning, match='gave up'):
                r = ses.push(f)
            assert eng.degraded
        else:
            r = ses.push(f)
        assert torch.isfinite(r['verts']).all()
        assert (r['verts'] - ref['verts'][j].cpu()).abs().max() < 2e-5      # step kernels vs persistent kernels: rounding
        assert (r['theta'] - ref['theta'][j].cpu()).abs().max() < 2e-5


def test_a_give_up_during_session_setup_is_repaired_too(smpl_np, monkeypatch):
    from tepose_amd.stream import StreamSession
    from tepose_amd.testing import build_model
    monkeypatch.setenv('TEPOSE_TEST_FAULT', '1')
    monkeypatch.setenv('TEPOSE_SEQ_SPIN_LIMIT', '20000')
    bad, _, _ = build_model(2, 256, seed=21, device='cuda', smpl_np=smpl_np)
    monkeypatch.delenv('TEPOSE_TEST_FAULT')
    monkeypatch.delenv('TEPOSE_SEQ_SPIN_LIMIT')
    w = synth.synthetic_windows(1, 9, 79)[0]
    with pytest.warns(RuntimeWarning, match='gave up'):
        ses = StreamSession(bad, 6, torch.from_numpy(w[:5, :2048].copy()), torch.from_numpy(w[:5, 2048:].copy()), keep=('theta',))
    assert bad._engine.degraded
    assert torch.isfinite(ses.push(torch.from_numpy(w[5, :2048].copy()))['theta']).all()


def test_the_session_owns_its_workspace_and_survives_other_callers_on_the_model(smpl_np):
    """the captured graph used to bake in the address of the engine's shared workspace, which a later, larger
    call on the same model frees and re-allocates (and which two sessions shared).  Now each session has a workspace of its
    own: between pushes the model runs a much larger batch (the shared workspace grows: other address), a second session on
    the SAME model interleaves its pushes, and a weight re-pack triggers a re-capture -- every frame still equals the
    clip-at-once driver bit for bit.  The session's buffers are initialised on a busy caller stream (ordering fix)."""
    from tepose_amd.driver import run_clips
    from tepose_amd.stream import StreamSession
    from tepose_amd.testing import build_model
    L, H, T, N = 2, 256, 6, 16
    model, _, _ = build_model(L, H, seed=31, device='cuda', smpl_np=smpl_np)
    wa, wb = synth.synthetic_windows(1, N, 91)[0], synth.synthetic_windows(1, N, 92)[0]
    fa, fb = torch.from_numpy(wa[:, :2048].copy()), torch.from_numpy(wb[:, :2048].copy())
    tha, thb = torch.from_numpy(wa[:T - 1, 2048:].copy()), torch.from_numpy(wb[:T - 1, 2048:].copy())
    ref_a = run_clips(model, [fa], [tha], T, keep=('theta', 'verts'))[0]
    ref_b = run_clips(model, [fb], [thb], T, keep=('theta', 'verts'))[0]
    # a busy default stream while the sessions are built from DEVICE tensors (the init copies must be ordered before capture)
    junk = torch.randn(4096, 4096, device='cuda')
    for _ in range(20):
        junk = junk @ junk.t() * 1e-4
    sa = StreamSession(model, T, fa[:T - 1].cuda(), tha.cuda(), keep=('theta', 'verts'))
    sb = StreamSession(model, T, fb[:T - 1].cuda(), thb.cuda(), keep=('theta', 'verts'))
    assert sa.ws.data_ptr() != sb.ws.data_ptr()
    eng = model._engine
    big = torch.from_numpy(synth.synthetic_windows(96, T, 93)).cuda()
    for j in range(N - T + 1):
        ra = {k: v.clone() for k, v in sa.push(fa[T - 1 + j]).items()}
        if j == 2:
            shared_before = eng._ws.data_ptr() if eng._ws is not None else 0
            with torch.no_grad():
                model(big)                                       # the shared workspace grows: freed + re-allocated
            assert eng._ws is not None and eng._ws.numel() >= eng.workspace_bytes(96, T)
            del shared_before
        if j == 4:
            gen = eng.packed_generation
            with torch.no_grad():
                model.encoder.linear_fwd.bias.add_(0.0)          # an in-place touch bumps the parameter version: re-pack
                model(big[:1])
            assert eng.packed_generation > gen
        rb = sb.push(fb[T - 1 + j])
        for k in ('theta', 'verts'):
            assert torch.equal(ra[k], ref_a[k][j].cpu()), (k, j)
            assert torch.equal(rb[k], ref_b[k][j].cpu()), (k, j)
    assert sa._captured_for[0] == eng.packed_generation        # both graphs were re-captured against the re-packed blob


@pytest.mark.parametrize('name', ['demo_L2H128_N24T6', 'demo_L1H64_N9T8'])
def test_the_demo_flow_of_the_reference_on_the_device(name, smpl_np):
    """BASELINE config 5 end to end as demo.py:209-262 runs it (the fixture is that block of the reference's file, executed): VIBE bootstrap over the
    tracklet -> first seq_len - 1 predictions = theta history -> sliding window with theta feedback, no J_regressor (49 joints), kp_2d.  Here: the drop-in
    VIBE, then the clip driver AND the frame-at-a-time StreamSession, every output of every frame within 1e-4 of the reference's."""
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    from test_gpu_vibe import _build as build_vibe
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    L, H, N, T, seed_w, seed_x = [int(v) for v in g['meta']]
    feats = torch.from_numpy(synth.synthetic_windows(1, N, seed_x)[:, :, :2048].copy())
    vibe, _ = build_vibe(L, H, seed_w + 1, smpl_np)
    with torch.no_grad():
        boot = vibe(feats.cuda())[-1]                                           # demo.py:229: model_vibe(batch)[-1], J_regressor = None
    for k, tol in (('theta', 1e-4), ('kp_3d', 1e-4), ('kp_2d', 1e-4)):
        got = boot[k].reshape(N, -1, g[k].shape[-1])[:T - 1].cpu().numpy() if k != 'theta' else boot[k].reshape(N, 85)[:T - 1].cpu().numpy()
        ref = g[k][:T - 1]
        if k == 'theta':                                                        # axis-angle near pi is compared through the outputs it produces
            assert np.abs(got[:, :3] - ref[:, :3]).max() < tol and np.abs(got[:, 75:] - ref[:, 75:]).max() < tol
        else:
            assert np.abs(got - ref).max() < tol, k
    assert np.abs(boot['verts'].reshape(N, 6890, 3)[:T - 1, ::53].cpu().numpy() - g['verts_sub'][:T - 1]).max() < 1e-4
    theta_init = boot['theta'].reshape(N, 85)[:T - 1].detach().clone()
    model, _, _ = build_model(L, H, seed=seed_w, device='cuda', smpl_np=smpl_np, seqlen=T)
    ref = run_clips(model, [feats[0].cuda()], [theta_init], T, keep=('theta', 'kp_3d', 'kp_2d', 'verts'))[0]
    for k in ('kp_3d', 'kp_2d'):
        assert np.abs(ref[k].cpu().numpy() - g[k][T - 1:]).max() < 1e-4, k
    assert np.abs(ref['verts'].cpu().numpy()[:, ::53] - g['verts_sub'][T - 1:]).max() < 1e-4
    th = ref['theta'].cpu().numpy()
    assert np.abs(th[:, :3] - g['theta'][T - 1:, :3]).max() < 1e-4 and np.abs(th[:, 75:] - g['theta'][T - 1:, 75:]).max() < 1e-4
    # frame at a time, as a live demo receives them: the same bits as the clip driver
    w = np.concatenate([feats[0].numpy(), np.zeros((N, 85), np.float32)], axis=1)
    got, ses = _stream_clip(model, w, theta_init.cpu().numpy(), T, None, keep=('theta', 'kp_3d', 'verts'))
    for k in ('theta', 'kp_3d', 'verts'):
        assert torch.equal(got[k], ref[k].cpu()), k


@pytest.mark.parametrize('graph', [True, False])
def test_an_in_place_weight_change_is_seen_by_the_next_push(graph, smpl_np):
    """parameters changed IN PLACE (an optimiser step, `p.add_()`, load_state_dict) with no eager forward in between -- the replayed graph would go
    on using the old packed blob without a sign.  push() compares the watched tensors' version counters (and re-walks the full signatures every 32nd
    push): the next frame runs on the new weights, graph and eager sessions alike, exactly as a model built with those weights from the start."""
    import copy
    from tepose_amd.stream import StreamSession
    from tepose_amd.testing import build_model
    T, N = 5, 14
    model, state, _ = build_model(1, 64, seed=41, device='cuda', smpl_np=smpl_np, seqlen=T)
    w = synth.synthetic_windows(1, N, 91)[0]
    feats, th0 = torch.from_numpy(w[:, :2048].copy()), torch.from_numpy(w[:T - 1, 2048:].copy())
    ses = StreamSession(model, T, feats[:T - 1], th0, keep=('theta', 'kp_3d'), graph=graph)
    for f in feats[T - 1:T + 2]:
        ses.push(f)
    gen0 = model._engine.packed_generation
    with torch.no_grad():
        model.regressor.deccam.bias.add_(0.05)                       # in place: same storage, version counter + 1
        model.encoder.gru_fwd.weight_hh_l0.mul_(0.9)
    got = [{k: v.clone() for k, v in ses.push(f).items()} for f in feats[T + 2:]]
    assert model._engine.packed_generation > gen0                     # re-packed by the push that followed the change
    # the same stream on a model that had the new weights from the start
    state2 = {k: np.array(v, copy=True) for k, v in state.items()}
    state2['regressor.deccam.bias'] = state2['regressor.deccam.bias'] + np.float32(0.05)
    state2['encoder.gru_fwd.weight_hh_l0'] = state2['encoder.gru_fwd.weight_hh_l0'] * np.float32(0.9)
    old, _, _ = build_model(1, 64, seed=41, device='cuda', smpl_np=smpl_np, seqlen=T)
    new, _, _ = build_model(1, 64, seed=41, device='cuda', smpl_np=smpl_np, seqlen=T, state=state2)
    ses_old = StreamSession(old, T, feats[:T - 1], th0, keep=('theta', 'kp_3d'), graph=graph)
    for f in feats[T - 1:T + 2]:
        ses_old.push(f)
    # hand the history the first session had reached over to the new-weights model: its window after the three pushes
    ses_new = StreamSession(new, T, feats[:T - 1], th0, keep=('theta', 'kp_3d'), graph=graph)
    ses_new.win.copy_(ses_old.win)
    want = [{k: v.clone() for k, v in ses_new.push(f).items()} for f in feats[T + 2:]]
    stale = [{k: v.clone() for k, v in ses_old.push(f).items()} for f in feats[T + 2:]]
    for a, b, c in zip(got, want, stale):
        assert torch.equal(a['theta'], b['theta']) and torch.equal(a['kp_3d'], b['kp_3d'])
        assert not torch.equal(a['theta'], c['theta'])                # ... and not what the old blob would have produced
