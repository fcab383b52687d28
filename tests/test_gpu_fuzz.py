"""GPU: bounded randomised parity sweep (the CI form of tests/_fuzz_parity.py; runs LAST with what is left of the suite's wall-clock budget, 6 .. 40 s:
tests/conftest.py): fixed seed, up to 90 random (L, H, B, T)
models / batches spanning every kernel family and dispatch threshold, HIP path vs the fp64 oracle -- encoder features in
both modes for every configuration, the full forward for B <= 300 -- plus a handful of H = 1024 cases at the batch sizes
BASELINE.json's configs use (64) and at the scaled-format thresholds (2048, 4096)."""
import time

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu


def _check(L, H, B, T, seed, smpl_np, J, full):
    from oracle import tepose_ref as O
    from tepose_amd.testing import build_model
    model, state, _ = build_model(L, H, seed=seed, device='cuda', smpl_np=smpl_np)
    x = synth.synthetic_windows(B, T, seed + 1)
    xd = torch.from_numpy(x).cuda()
    with torch.no_grad():
        f = model.encoder(xd)
        ft = model.encoder(xd, is_train=True)
        out = model(xd, J_regressor=J)[0] if full else None
    enc, _ = O.split_state_dict(state, torch.float64)
    # windows are independent rows: for big batches of wide models the fp64 oracle runs on the first and the last 128-row tile's edge rows and on rows
    # drawn from the rest (every kernel treats all row tiles alike); the whole batch is checked for finiteness
    rows = np.arange(B) if B * H <= 150000 else np.unique(np.r_[0:24, B - 24:B, np.random.RandomState(B).randint(0, B, 48)])
    with torch.no_grad():
        rf = O.encoder_fwd(enc, torch.from_numpy(x[rows]).double(), L)
        rft = O.encoder_fwd(enc, torch.from_numpy(x[rows]).double(), L, is_train=True)
    assert torch.isfinite(f).all() and torch.isfinite(ft).all()
    e1 = (f.cpu().double()[rows] - rf).abs().max().item()
    e2 = (ft.cpu().double()[rows] - rft).abs().max().item()
    e3 = 0.0
    if out is not None:
        ref = O.tepose_fwd(state, smpl_np, x, L, J_regressor=smpl_np['J_regressor_h36m'], dtype=torch.float64)
        e3 = max((out[k].cpu().double() - ref[k]).abs().max().item() for k in ('verts', 'kp_3d', 'rotmat', 'kp_2d'))
    del model
    return e1, e2, e3


def test_random_configurations_against_fp64_oracle(fuzz_budget_s):
    rng = np.random.RandomState(20261003)
    smpl_np = synth.synthetic_smpl(0)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    t0 = time.time()
    worst, n = 0.0, 0
    Bs = [1, 2, 3, 4, 5, 7, 16, 17, 31, 33, 48, 64, 65, 100, 128, 129, 200, 257, 400, 640, 769, 1000, 1280, 2048, 2100, 4096]   # % 128 == 0 from 640: the plane-fed step kernel
    for i in range(90):
        L = int(rng.choice([1, 2, 2, 3]))
        H = int(rng.choice([64, 100, 128, 192, 256, 320]))
        B = int(Bs[i % len(Bs)] if i < len(Bs) else rng.choice(Bs))       # every batch class at least once
        T = int(rng.choice([1, 2, 3, 5, 6, 8]))
        seed = int(rng.randint(1 << 20))
        e1, e2, e3 = _check(L, H, B, T, seed, smpl_np, J, full=B <= 300)
        assert max(e1, e2) < 2e-5 and e3 < 1e-4, (L, H, B, T, seed, e1, e2, e3)
        worst, n = max(worst, e1, e2, e3), n + 1
        if time.time() - t0 > fuzz_budget_s:            # bounded by what is left of the suite's wall-clock budget (tests/conftest.py: 6 .. 40 s; runs last);
            break                                       # the sweep order is deterministic: a shorter budget runs a prefix
    assert n >= 6, n
    print('fuzz: %d configurations, worst abs error %.2e, %.0f s of a %.0f s budget' % (n, worst, time.time() - t0, fuzz_budget_s))


@pytest.mark.parametrize('B,T', [(64, 16), (2048, 2), (37, 6), (128, 4), (640, 3)])
def test_published_architecture_batches_against_fp64_oracle(B, T):
    """n_layers = 2, hidden = 1024 (the published checkpoints): BASELINE.json config 2's shape (B = 64, T = 16), the
    37-clip lock-step shape of the 3DPW-test evaluation, and the batch thresholds of the large-batch kernels."""
    smpl_np = synth.synthetic_smpl(0)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    e1, e2, e3 = _check(2, 1024, B, T, 5, smpl_np, J, full=B <= 128)
    assert max(e1, e2) < 2e-5 and e3 < 1e-4, (B, T, e1, e2, e3)
