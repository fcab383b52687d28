"""Autoregressive sliding-window driver, batched over clips.

The reference runs TePose one window at a time per clip: window j's predicted theta is
written into the theta slots of the following windows (evaluate.py:247-269, demo.py:238-252;
batched-clip form lib/core/trainer.py:313-344).  Windows of one clip are therefore serial;
clips are independent, so all clips advance in lock-step and the model sees B = number of
clips still running.  Clips are processed longest-first, which keeps the active set a prefix.

Host side is tensor plumbing only (slice / copy on the device); every window step is one
`TePose.forward` = one call into libtepose_hip.so.
"""
import torch


@torch.no_grad()
def run_clips(model, features, theta_init, seqlen, J_regressor=None, keep=('theta', 'kp_3d', 'verts', 'rotmat')):
    """features: list of [N_i, 2048] tensors (one per clip); theta_init: list of [seqlen-1, 85]
    (theta of the first seqlen-1 frames: pseudo-theta with cam=[1,0,0] in evaluate.py:177,219,
    VIBE output in demo.py:237).  Clips shorter than seqlen are skipped (evaluate.py:226-227).

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
    F = torch.zeros(C, nmax, 2048, device=dev)
    TH = torch.zeros(C, nmax, 85, device=dev)
    for s, i in enumerate(order):
        F[s, :n[i]] = features[i].to(dev, torch.float32)
        TH[s, :T - 1] = theta_init[i].to(dev, torch.float32)
    steps = [n[i] - T + 1 for i in order]                 # windows per clip, non-increasing
    outs = {k: [None] * C for k in keep}
    bufs = {}
    inp = torch.zeros(C, T, 2133, device=dev)
    for j in range(steps[0]):
        b = sum(1 for s in steps if s > j)                # active clips form the prefix [0, b)
        x = inp[:b]
        x[:, :, :2048] = F[:b, j:j + T]
        x[:, :T - 1, 2048:] = TH[:b, j:j + T - 1]         # last frame's theta stays zero
        pred = model(x, J_regressor=J_regressor)[0]
        TH[:b, j + T - 1] = pred['theta']                 # feeds the next windows
        for k in keep:
            if k not in bufs:
                bufs[k] = torch.empty((C, steps[0]) + tuple(pred[k].shape[1:]), device=dev)
            bufs[k][:b, j] = pred[k]
    for s, i in enumerate(order):
        results[i] = {k: bufs[k][s, :steps[s]].clone() for k in keep}
    return results
