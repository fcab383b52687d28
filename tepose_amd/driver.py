"""Autoregressive sliding-window driver, batched over clips.

The reference runs TePose one window at a time per clip: window j's predicted theta is
written into the theta slots of the following windows (evaluate.py:247-269, demo.py:238-252;
batched-clip form lib/core/trainer.py:313-344).  Windows of one clip are therefore serial;
clips are independent, so all clips advance in lock-step and the model sees B = number of
clips still running.  Clips are processed longest-first, which keeps the active set a prefix.

Host side is tensor plumbing only (slice / copy on the device); every window step is one
`TePose.forward` = one call into libtepose_hip.so.
"""
import os

import torch

from . import _lib
from .engine import on_device


@torch.no_grad()
def run_clips(model, features, theta_init, seqlen, J_regressor=None, keep=('theta', 'kp_3d', 'verts', 'rotmat'),
              cache_projections='auto'):
    """`_run_clips` under the failure contract of the persistent small-batch kernels (include/tepose_amd.h, "failure
    channel"): the window steps are queued without a host sync (`lazy` status mode), the handle's fault word is read at
    the end -- the driver's one sync point -- and a run during which a kernel gave up is repeated on the step-per-launch
    HIP kernels (the engine has switched the handle over and warned by then).  Never NaN results, never a CPU path."""
    eng = model._engine
    for attempt in (0, 1):
        try:
            with eng.lazy_status():
                res = _run_clips(model, features, theta_init, seqlen, J_regressor, keep, cache_projections)
                eng.check_status()
            return res
        except _lib.TeposeTimeout:
            if attempt:
                raise


def _run_clips(model, features, theta_init, seqlen, J_regressor, keep, cache_projections):
    """features: list of [N_i, 2048] tensors (one per clip); theta_init: list of [seqlen-1, 85]
    (theta of the first seqlen-1 frames: pseudo-theta with cam=[1,0,0] in evaluate.py:177,219,
    VIBE output in demo.py:237).  Clips shorter than seqlen are skipped (evaluate.py:226-227).

    cache_projections: keep every frame's layer-0 gate pre-activations (x W_ih^T + b_ih, 3 directions) in a
    per-clip ring, so a window step projects 2 frames per clip (the newest one with zero theta and the
    previous one with its now-known theta) instead of all `seqlen` (saves up to 42 % of the
    FLOPs).  'auto' turns it on from 2 concurrent clips.

    Returns a list (same order as the input) of dicts key -> [N_i - seqlen + 1, ...] tensors:
    the prediction for the last frame of every window, i.e. frames seqlen-1 .. N_i-1."""
    T = int(seqlen)
    dev = next(model.parameters()).device
    n = [int(f.shape[0]) for f in features]
    order = sorted([i for i in range(len(features)) if n[i] >= T], key=lambda i: (-n[i], i))
    results = [None] * len(features)
    if not order:
        return results
    C, nmax = len(order), n[order[0]]
    nj = 14 if J_regressor is not None else 49
    tails = {'theta': (85,), 'verts': (6890, 3), 'kp_2d': (nj, 2), 'kp_3d': (nj, 3), 'rotmat': (24, 3, 3)}
    F = torch.zeros(C, nmax, 2048, device=dev)
    # theta history FRAME-major [frame][clip][85]: the forward of window j writes its predictions straight into frame j + T - 1 (rows [0, b) = the active
    # clips: a prefix), where the next windows read them -- and the same buffer is the result `theta`.  Every kept output likewise has a step-major buffer
    # [step][clip][...] whose slice [j, :b] is contiguous: a window step costs NO device copy and no tensor op besides the forward itself.
    TH = torch.zeros(nmax, C, 85, device=dev)
    for s, i in enumerate(order):
        F[s, :n[i]] = features[i].to(dev, torch.float32)
        TH[:T - 1, s] = theta_init[i].to(dev, torch.float32)
    steps = [n[i] - T + 1 for i in order]                 # windows per clip, non-increasing
    bufs = {k: torch.empty((steps[0], C) + tails[k], device=dev) for k in keep if k != 'theta'}
    eng = model._engine
    # measured (tools/driver_cache_crossover.py, round 5): with the step as one library call the cache wins at every clip count (1 clip: -3 ... -5 %);
    # one clip stays uncached so that it runs the very kernels of the live-stream session (tepose_amd.stream, compared bit for bit in the tests)
    use_cache = (C >= 2) if cache_projections == 'auto' else bool(cache_projections)
    if cache_projections == 'auto' and os.environ.get('TEPOSE_DRIVER_CACHE', '') in ('0', '1'):    # A/B and debugging
        use_cache = os.environ['TEPOSE_DRIVER_CACHE'] == '1'
    with on_device(dev):
        eng.pack_encoder(model.encoder, dev)
        eng.pack_regressor(model.regressor, dev)
    if use_cache:
        ring_n = max(T - 1, 1)
        ring = torch.empty(C, ring_n, eng.gate_width, device=dev)
        newest = torch.empty(C, eng.gate_width, device=dev)
        pws = torch.empty(int(eng.lib.tepose_project_frames_workspace_bytes(eng.handle, 2 * C)), dtype=torch.uint8, device=dev)
        th_ld = TH.stride(1)
        pair = os.environ.get('TEPOSE_DRIVER_PAIR', '1') != '0'      # A/B: 0 = the two projections of a step as two products
        one_call = pair and os.environ.get('TEPOSE_DRIVER_PAIR', '1') != '2'      # A/B: 2 = pair projection and forward as two library calls

        def project(frame, theta, b):
            out = newest if theta is None else ring[:, frame % ring_n]
            eng.project_frames(F[:, frame].data_ptr(), F.stride(0), None if theta is None else TH[frame].data_ptr(),
                               th_ld, b, out.data_ptr(), out.stride(0), pws)
        with on_device(dev):
            for t in range(T - 1):
                project(t, True, C)
    else:
        inp = torch.zeros(C, T, 2133, device=dev)
    for j in range(steps[0]):
        b = sum(1 for s in steps if s > j)                # active clips form the prefix [0, b)
        out = {k: v[j] for k, v in bufs.items()}
        out['theta'] = TH[j + T - 1]                      # feeds the next windows
        with on_device(dev):
            if use_cache:
                if j > 0 and not pair:
                    project(j + T - 2, True, b)
                    project(j + T - 1, None, b)
                elif j > 0 and one_call:
                    # the whole step as ONE library call: both projections as one product of 2 b rows, then the forward from the cached projections
                    fp, fn = j + T - 2, j + T - 1
                    eng.window_step(F[:, fp].data_ptr(), F[:, fn].data_ptr(), F.stride(0), TH[fp].data_ptr(), th_ld, ring[:, fp % ring_n].data_ptr(),
                                    ring.stride(0), ring, j % ring_n, newest, b, T, J_regressor, pws, out=out)
                    continue
                elif j > 0:
                    # previous newest frame (theta now known) -> its ring slot, newest frame (theta slots zero) -> `newest`: ONE product of 2 b rows
                    fp, fn = j + T - 2, j + T - 1
                    eng.project_frame_pair(F[:, fp].data_ptr(), F[:, fn].data_ptr(), F.stride(0), TH[fp].data_ptr(), th_ld, b,
                                           ring[:, fp % ring_n].data_ptr(), ring.stride(0), newest.data_ptr(), newest.stride(0), pws)
                else:
                    project(j + T - 1, None, b)           # newest frame, theta slots zero
                eng.forward_cached(ring, j % ring_n, newest, b, T, J_regressor, out=out)
            else:
                x = inp[:b]
                x[:, :, :2048] = F[:b, j:j + T]
                x[:, :T - 1, 2048:] = TH[j:j + T - 1, :b].transpose(0, 1)     # last frame's theta stays zero
                eng.forward(x, J_regressor, out=out)
    for s, i in enumerate(order):
        results[i] = {k: (TH[T - 1:T - 1 + steps[s], s] if k == 'theta' else bufs[k][:steps[s], s]).clone() for k in keep}
    return results


@torch.no_grad()
def validate_padded(model, target, seqlen, J_regressor=None, keep=('theta', 'kp_3d', 'verts')):
    """The batched whole-clip validation loop of the reference's trainer (lib/core/trainer.py:307-357) on one batch of its
    validation Datasets (lib/dataset/threedpw_test.py:62-134, h36m_val.py): target['features'] [C, vidlen, 2048] (clips
    zero-padded to the longest), target['theta_pseu'] [C, vidlen, 85] (cam = [1, 0, 0]), target['vidlen_each'] [C, 1].

    The reference advances every clip through all vidlen - seqlen + 1 windows, padding included, and keeps the rows with
    j < vidlen_each - seqlen + 1; here a clip stops at its own last window (run_clips: the active clips are a prefix of the
    longest-first order), which yields the same kept rows.  Returns the trainer's accumulators in its order -- for j in
    windows, the clips still running, in batch order: 'pred_<key>' [sum_c (vidlen_c - seqlen + 1), ...] for key in `keep`
    -- and 'pred_j3d_tsr' [C, vidlen, J, 3] with the prediction for frame j + seqlen - 1 of clip c (zeros in front of the
    first window and on the padding, where the reference holds predictions from zero features)."""
    T = int(seqlen)
    feats, th = target['features'], target['theta_pseu']
    C, vidlen = int(feats.shape[0]), int(feats.shape[1])
    lens = [int(v) for v in target['vidlen_each'].reshape(-1).tolist()]
    res = run_clips(model, [feats[c, :lens[c]] for c in range(C)], [th[c, :T - 1] for c in range(C)], T,
                    J_regressor=J_regressor, keep=tuple(keep))
    nwin = [max(lens[c] - T + 1, 0) if res[c] is not None else 0 for c in range(C)]
    out = {}
    order = [(j, c) for j in range(max(nwin + [0])) for c in range(C) if j < nwin[c]]
    for k in keep:
        rows = [res[c][k][j] for j, c in order]
        out['pred_' + k] = torch.stack(rows) if rows else None
    if 'kp_3d' in keep and order:
        nj = int(res[order[0][1]]['kp_3d'].shape[1])
        tsr = torch.zeros((C, vidlen, nj, 3), device=out['pred_kp_3d'].device)
        for c in range(C):
            if nwin[c]:
                tsr[c, T - 1:T - 1 + nwin[c]] = res[c]['kp_3d']
        out['pred_j3d_tsr'] = tsr
    return out
