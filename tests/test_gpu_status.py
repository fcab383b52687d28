"""GPU: the failure channel of the persistent small-batch kernels (include/tepose_amd.h "failure channel";
csrc/gru_seq.hip, csrc/reg_seq.hip).  A launch whose workgroups cannot all be resident (GPU shared / CU-masked) gives up
after a bounded wait.  The reference has no silent-garbage mode -- an exception ends evaluate.py:255 -- so here a give-up
must surface as TEPOSE_E_TIMEOUT / TeposeTimeout or be repaired by a re-run on the step-per-launch HIP kernels; the
caller never receives the NaN outputs with rc 0.

The give-up is forced with the test-only knob TEPOSE_TEST_FAULT (read at tepose_create: the waits of that handle's
persistent kernels expect one more arrival / tag than anybody publishes) and a short TEPOSE_SEQ_SPIN_LIMIT."""
import ctypes
import warnings

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def smpl_np():
    return synth.synthetic_smpl(0)


def _faulty_model(monkeypatch, smpl_np, bits, L=2, H=256, seed=21, mode=None):
    from tepose_amd.testing import build_model
    monkeypatch.setenv('TEPOSE_TEST_FAULT', str(bits))
    monkeypatch.setenv('TEPOSE_SEQ_SPIN_LIMIT', '20000')          # ~20 ms instead of ~2 s per give-up
    if mode:
        monkeypatch.setenv('TEPOSE_STATUS_CHECK', mode)
    model, state, _ = build_model(L, H, seed=seed, device='cuda', smpl_np=smpl_np)
    monkeypatch.delenv('TEPOSE_TEST_FAULT')
    monkeypatch.delenv('TEPOSE_SEQ_SPIN_LIMIT')
    return model, state


def _oracle(state, smpl_np, x, L, J):
    from oracle import tepose_ref as O
    return O.tepose_fwd(state, smpl_np, x, L, J_regressor=J)


@pytest.mark.parametrize('B', [1, 3, 8, 40])         # granule mode (<= 4 rows) and counter mode, 1 / 4 row tiles
def test_give_up_is_repaired_in_sync_mode(B, smpl_np, monkeypatch):
    """Default status mode: model(x) notices the give-up before it returns, warns, re-runs on the step kernels and hands
    out correct results."""
    model, state = _faulty_model(monkeypatch, smpl_np, 1)
    assert model._engine.status_mode == 'sync'
    x = synth.synthetic_windows(B, 6, 9)
    J = smpl_np['J_regressor_h36m']
    with pytest.warns(RuntimeWarning, match='gave up'):
        with torch.no_grad():
            out = model(torch.from_numpy(x).cuda(), J_regressor=torch.from_numpy(J))[0]
    assert model._engine.degraded
    ref = _oracle(state, smpl_np, x, 2, J)
    for k in ('theta', 'verts', 'kp_3d'):
        got = out[k].cpu().numpy()
        assert np.isfinite(got).all(), k
        assert np.abs(got - ref[k].numpy()).max() < 1e-4, k
    # the handle stays on the step kernels: no further warning, same results
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        with torch.no_grad():
            out2 = model(torch.from_numpy(x).cuda(), J_regressor=torch.from_numpy(J))[0]
    assert torch.equal(out2['verts'], out['verts'])


def test_give_up_raises_in_lazy_mode_and_at_the_c_boundary(smpl_np, monkeypatch):
    """Lazy mode (callers that own their sync points): the forward itself returns, the fault surfaces at check_status()
    as TeposeTimeout; at the C boundary tepose_status_peek / tepose_status return TEPOSE_E_TIMEOUT and a forward entered
    while the word is raised is refused with the same code."""
    from tepose_amd import _lib
    model, state = _faulty_model(monkeypatch, smpl_np, 1, mode='lazy')
    eng = model._engine
    assert eng.status_mode == 'lazy'
    x = torch.from_numpy(synth.synthetic_windows(2, 5, 10)).cuda()
    with torch.no_grad():
        out = model(x)[0]
    torch.cuda.synchronize()
    assert not torch.isfinite(out['verts']).all()        # what the boundary must never hand over silently
    lib = eng.lib
    assert lib.tepose_status_peek(eng.handle) == _lib.E_TIMEOUT
    # a second forward is refused up front (rc, not NaN), the handle switches kernels, the exception reaches the caller
    with pytest.warns(RuntimeWarning, match='gave up'):
        with pytest.raises(_lib.TeposeTimeout):
            with torch.no_grad():
                model(x)
    assert lib.tepose_status_peek(eng.handle) == 0
    with torch.no_grad():
        good = model(x)[0]
    eng.check_status()                                    # nothing pending: no exception
    assert torch.isfinite(good['verts']).all()
    assert b'bounded wait' in lib.tepose_error_string(_lib.E_TIMEOUT)


def test_check_status_raises_after_a_faulted_forward(smpl_np, monkeypatch):
    from tepose_amd import _lib
    model, _ = _faulty_model(monkeypatch, smpl_np, 1, mode='lazy')
    x = torch.from_numpy(synth.synthetic_windows(5, 4, 11)).cuda()
    with torch.no_grad():
        model(x)
    with pytest.warns(RuntimeWarning, match='gave up'):
        with pytest.raises(_lib.TeposeTimeout):
            model._engine.check_status()
    assert model._engine.degraded and not model._engine.uses_persistent(5)


def test_regressor_kernel_give_up(smpl_np, monkeypatch):
    """The persistent FC-loop kernel (n_iter != 3 keeps the loop): same contract."""
    from oracle import tepose_ref as O
    model, state = _faulty_model(monkeypatch, smpl_np, 2)
    feat = synth.normal('status/feat', (6, 2048), std=0.5)
    with pytest.warns(RuntimeWarning, match='gave up'):
        with torch.no_grad():
            out = model.regressor(torch.from_numpy(feat).cuda(), n_iter=2)[0]
    _, reg = O.split_state_dict(state, torch.float64)
    with torch.no_grad():
        ref = O.regressor_fwd(reg, O.smpl_tensors(smpl_np, torch.float64), torch.from_numpy(feat).double(), n_iter=2)
    assert np.abs(out['verts'].cpu().numpy() - ref['verts'].numpy()).max() < 1e-4


def test_driver_repeats_a_faulted_run(smpl_np, monkeypatch):
    """run_clips queues its window steps without syncing; a give-up anywhere in the run is read at the end and the run
    is repeated on the step kernels: same results as a healthy handle."""
    from tepose_amd.driver import run_clips
    from tepose_amd.testing import build_model
    good, _, _ = build_model(2, 256, seed=21, device='cuda', smpl_np=smpl_np)
    bad, _ = _faulty_model(monkeypatch, smpl_np, 1)
    feats = [torch.from_numpy(synth.normal('status/clip%d' % i, (n, 2048), std=0.5)) for i, n in enumerate((9, 7))]
    th0 = [torch.from_numpy(synth.normal('status/th%d' % i, (3, 85), std=0.1)) for i in range(2)]
    ref = run_clips(good, feats, th0, 4)
    with pytest.warns(RuntimeWarning, match='gave up'):
        got = run_clips(bad, feats, th0, 4)
    for r, g in zip(ref, got):
        assert torch.isfinite(g['verts']).all()
        assert (r['verts'] - g['verts']).abs().max() < 2e-5      # persistent vs step kernels: K-partials grouped differently
        assert (r['theta'] - g['theta']).abs().max() < 2e-5


def test_persistent_kernels_can_be_switched_off_up_front(smpl_np, monkeypatch):
    """TEPOSE_PERSISTENT=0: the documented start-up remedy for shared / CU-masked GPUs."""
    from tepose_amd.testing import build_model
    monkeypatch.setenv('TEPOSE_PERSISTENT', '0')
    monkeypatch.setenv('TEPOSE_TEST_FAULT', '3')                  # would fault if a persistent kernel ran
    model, state, _ = build_model(2, 256, seed=21, device='cuda', smpl_np=smpl_np)
    x = synth.synthetic_windows(3, 5, 12)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        with torch.no_grad():
            out = model(torch.from_numpy(x).cuda())[0]
    assert not model._engine.uses_persistent(3)
    ref = _oracle(state, smpl_np, x, 2, None)
    assert np.abs(out['verts'].cpu().numpy() - ref['verts'].numpy()).max() < 1e-4


def test_fault_is_attributed_to_the_forward_not_the_handle(smpl_np, monkeypatch):
    """One handle, two streams, two workspaces (the re-entrancy the C header documents).  The forward on stream A gives up; the
    forward on stream B -- a batch that launches no persistent kernel -- runs concurrently and is healthy.
    tepose_forward_status answers per forward (the status words live in the workspace): B's caller, polling FIRST, gets 0 and
    does not consume A's fault; A's caller gets TEPOSE_E_TIMEOUT exactly once.  (The handle-wide tepose_status would have handed
    A's fault to whoever polls first: ADVICE r03.)  Engine.set_persistent(True) re-arms a degraded model."""
    from tepose_amd import _lib
    model, state = _faulty_model(monkeypatch, smpl_np, 1, mode='lazy')
    eng, lib = model._engine, model._engine.lib
    T, BA, BB = 5, 3, 200
    xa = torch.from_numpy(synth.synthetic_windows(BA, T, 14)).cuda()
    xb = torch.from_numpy(synth.synthetic_windows(BB, T, 15)).cuda()
    with torch.no_grad():
        model(xb)                                                   # packs; B = 200 runs no persistent kernel, so no fault
    eng.check_status()
    assert eng.uses_persistent(BA) and not eng.uses_persistent(BB)

    def buffers(B):
        ws = torch.empty(int(lib.tepose_workspace_bytes(eng.handle, B, T)), dtype=torch.uint8, device='cuda')
        return ws, eng._outputs(B, 49, xa.device)

    wsa, oa = buffers(BA)
    wsb, ob = buffers(BB)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()

    def fwd(x, B, ws, o, st):
        return lib.tepose_forward(eng.handle, x.data_ptr(), B, T, None, o['theta'].data_ptr(), o['verts'].data_ptr(), o['kp_3d'].data_ptr(),
                                  o['kp_2d'].data_ptr(), o['rotmat'].data_ptr(), ws.data_ptr(), ws.numel(), st.cuda_stream)
    assert fwd(xb, BB, wsb, ob, sb) == 0                            # healthy; queued first (an entry point refuses once the word is up)
    assert fwd(xa, BA, wsa, oa, sa) == 0                            # gives up after its bounded wait (~20 ms), concurrently with B
    assert lib.tepose_forward_status(eng.handle, wsb.data_ptr(), sb.cuda_stream) == 0          # B polls first ...
    assert torch.isfinite(ob['verts']).all()
    assert lib.tepose_forward_status(eng.handle, wsa.data_ptr(), sa.cuda_stream) == _lib.E_TIMEOUT   # ... A's fault is still A's
    assert not torch.isfinite(oa['verts']).all()
    assert lib.tepose_forward_status(eng.handle, wsa.data_ptr(), sa.cuda_stream) == 0          # once
    assert lib.tepose_status_peek(eng.handle) == 0                  # and the handle is usable again
    ref = _oracle(state, smpl_np, xb.cpu().numpy()[:8], 2, None)
    assert np.abs(ob['verts'][:8].cpu().numpy() - ref['verts'].numpy()).max() < 1e-4
    # the remedy and the way back
    eng.set_persistent(False)
    assert not eng.uses_persistent(BA)
    assert fwd(xa, BA, wsa, oa, sa) == 0 and lib.tepose_forward_status(eng.handle, wsa.data_ptr(), sa.cuda_stream) == 0
    assert torch.isfinite(oa['verts']).all()
    eng.set_persistent(True)
    assert eng.uses_persistent(BA) and not eng.degraded


def test_barrier_free_projection_give_up_reaches_the_failure_channel(smpl_np, monkeypatch):
    """csrc/gemm_h3s16c.hip has no workgroup barriers: its waves wait on LDS arrival counters with BOUNDED polls.  A poll that
    expires (only a kernel bug or a hardware fault can make it: there is no dependence between workgroups) must not end as NaN
    with rc 0 either: the kernel raises the forward's status word and the handle's fault word (TEPOSE_TEST_FAULT bit 2 adds one
    to every poll target).  Large batches are never synchronised by the library, so the channel is the lazy one: the NEXT entry
    point refuses with TEPOSE_E_TIMEOUT, tepose_forward_status on the forward's workspace says so once."""
    from tepose_amd import _lib
    model, state = _faulty_model(monkeypatch, smpl_np, 4, mode='lazy')
    eng, lib = model._engine, model._engine.lib
    B, T = 600, 16                                                 # 9600 rows: the layer-0 projection runs on the persistent 256 x 256-tile kernel
    x = torch.from_numpy(synth.synthetic_windows(B, T, 31)).cuda()
    assert 'persist16c' in eng.kernel_info()['projection']
    errs0 = int(lib.tepose_debug_kernel_errors())
    with torch.no_grad():
        feat = model.encoder(x)                                     # rc 0: nothing has been synchronised yet
    torch.cuda.synchronize()
    assert int(lib.tepose_debug_kernel_errors()) > errs0           # the kernel's own counter
    assert lib.tepose_status_peek(eng.handle) != 0                 # the handle's host-visible fault word is up
    assert lib.tepose_fault_code(eng.handle) == 4                  # ... and says WHICH kernel: the barrier-free projection, not a residency problem
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tepose_forward_status(eng.handle, eng._ws.data_ptr(), st) == _lib.E_TIMEOUT       # the forward's own words: reported once
    assert lib.tepose_forward_status(eng.handle, eng._ws.data_ptr(), st) == 0
    assert lib.tepose_fault_code(eng.handle) == 4 and lib.tepose_status_peek(eng.handle) == 0
    # the same through the engine: the next entry point refuses before queueing more work, as a DISTINCT exception -- the handle is not switched to
    # the step-per-launch kernels and no shared-GPU warning is printed (code 4 has nothing to do with the persistent small-batch kernels)
    with torch.no_grad():
        model.encoder(x)
    torch.cuda.synchronize()
    assert lib.tepose_status_peek(eng.handle) != 0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        with pytest.raises(_lib.TeposeKernelFault):
            with torch.no_grad():
                model.encoder(x)
    assert not eng.degraded and eng.uses_persistent(1)
    # the refusal collected the HANDLE's word; the forward's own status words still answer for that forward (a handle-wide collection -- by
    # this caller or any other thread -- must not hide a give-up from the forward's own check), once
    assert lib.tepose_forward_status(eng.handle, eng._ws.data_ptr(), st) == _lib.E_TIMEOUT
    assert lib.tepose_forward_status(eng.handle, eng._ws.data_ptr(), st) == 0
    # without the injected fault the same handle is healthy again
    assert lib.tepose_debug_set_test_fault(eng.handle, 0) == 0
    eng.check_status()
    with torch.no_grad():
        good = model.encoder(x)
    torch.cuda.synchronize()
    assert lib.tepose_status_peek(eng.handle) == 0 and torch.isfinite(good).all()
    from oracle import tepose_ref as O
    enc, _ = O.split_state_dict(state, torch.float64)
    with torch.no_grad():
        ref = O.encoder_fwd(enc, x[:64].cpu().double(), 2)
    assert (good[:64].cpu().double() - ref).abs().max() < 2e-5
    del feat


def test_a_fault_collected_by_another_caller_is_still_reported_to_its_forward(smpl_np, monkeypatch):
    """the handle's fault word is shared by every stream and thread and ANY status call clears it.  Forward A gives up; before
    A's caller asks, another caller collects the word handle-wide (tepose_status).  A's own tepose_forward_status must not take the clear
    word for "nobody gave up": the library counts collections and falls back to A's status words -- TEPOSE_E_TIMEOUT, exactly once."""
    from tepose_amd import _lib
    model, _ = _faulty_model(monkeypatch, smpl_np, 1, mode='lazy')
    eng, lib = model._engine, model._engine.lib
    x = torch.from_numpy(synth.synthetic_windows(2, 5, 16)).cuda()
    with torch.no_grad():
        out = model(x)[0]                                           # lazy mode: queued, gives up, nobody has looked yet
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tepose_status(eng.handle, st) == _lib.E_TIMEOUT      # the other caller: syncs, collects, clears
    assert lib.tepose_status_peek(eng.handle) == 0
    assert not torch.isfinite(out['verts']).all()
    assert lib.tepose_forward_status(eng.handle, eng._ws.data_ptr(), st) == _lib.E_TIMEOUT
    assert lib.tepose_forward_status(eng.handle, eng._ws.data_ptr(), st) == 0


def test_sync_mode_never_returns_nan_while_another_thread_polls_the_handle(smpl_np, monkeypatch):
    """The same race with real threads: a poller calls tepose_status on a side stream in a loop while the main thread runs faulting
    forwards in the default (sync) mode, re-arming the persistent kernels every time.  Every forward must come back repaired (finite,
    equal to the step-per-launch result) although the poller regularly gets to the fault word first."""
    import threading
    import time
    model, _ = _faulty_model(monkeypatch, smpl_np, 1)
    eng, lib = model._engine, model._engine.lib
    x = torch.from_numpy(synth.synthetic_windows(3, 5, 17)).cuda()
    side = torch.cuda.Stream()
    stop, stolen = threading.Event(), [0]

    def poll():
        while not stop.is_set():
            if lib.tepose_status(eng.handle, side.cuda_stream) != 0:
                stolen[0] += 1
            time.sleep(0.0005)

    th = threading.Thread(target=poll)
    th.start()
    try:
        ref = None
        for it in range(12):
            eng.set_persistent(True)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                try:
                    with torch.no_grad():
                        out = model(x)[0]
                except Exception as e:                              # refused up front because the poller has not collected yet: fine, never silent
                    from tepose_amd import _lib
                    assert isinstance(e, _lib.TeposeTimeout)
                    continue
            torch.cuda.synchronize()
            assert torch.isfinite(out['verts']).all() and torch.isfinite(out['theta']).all(), it
            if ref is None:
                ref = out['verts'].clone()
            assert torch.equal(out['verts'], ref), it
    finally:
        stop.set()
        th.join()
    assert ref is not None
    print('  the poller collected the fault word %d times during the 12 forwards' % stolen[0])
