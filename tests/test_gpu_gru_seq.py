"""GPU: the persistent recurrent kernel (csrc/gru_seq.hip: all T cell steps of a layer in one launch, W_hh stationary
in registers, per-direction arrival counters between steps) against the fp64 oracle and against the step-per-launch
kernels it replaces -- at every tile configuration (1 / 2 / 4 row tiles, 1..4 K-tiles per wave), ragged batch sizes,
the longest supported window, under repetition and next to a competing stream (the hand-off must not depend on timing)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def smpl_np():
    return synth.synthetic_smpl(0)


def _model(L, H, seed, smpl_np):
    from tepose_amd.testing import build_model
    return build_model(L, H, seed=seed, device='cuda', smpl_np=smpl_np)


# (L, H, B, T): H % 256 == 0 and B <= 64 take the persistent kernel (B <= 4: states handed over as tagged granules)
SHAPES = [(2, 1024, 1, 16), (2, 1024, 3, 32), (2, 1024, 4, 6), (1, 512, 1, 7), (3, 256, 4, 3), (2, 512, 2, 36),
          (2, 1024, 5, 6), (2, 1024, 64, 16), (2, 1024, 37, 6), (2, 1024, 16, 16), (2, 1024, 33, 2),
          (2, 256, 17, 5), (1, 512, 33, 4), (3, 256, 64, 3), (2, 768, 16, 7), (1, 1024, 48, 5), (2, 1024, 8, 36), (2, 256, 49, 3)]


@pytest.mark.parametrize('L,H,B,T', SHAPES)
def test_encoder_on_persistent_kernel_vs_oracle(L, H, B, T, smpl_np):
    from oracle import tepose_ref as O
    model, state, _ = _model(L, H, 31, smpl_np)
    x = synth.synthetic_windows(B, T, 77)
    xd = torch.from_numpy(x).cuda()
    with torch.no_grad():
        feat = model.encoder(xd)
        feat_tr = model.encoder(xd, is_train=True)
    enc, _ = O.split_state_dict(state, torch.float64)
    with torch.no_grad():
        ref = O.encoder_fwd(enc, torch.from_numpy(x).double(), L)
        ref_tr = O.encoder_fwd(enc, torch.from_numpy(x).double(), L, is_train=True)
    assert (feat.cpu().double() - ref).abs().max() < 2e-5
    assert (feat_tr.cpu().double() - ref_tr).abs().max() < 2e-5


@pytest.mark.parametrize('B', [64, 1, 3])
def test_repeated_forwards_are_identical_under_a_competing_stream(B, smpl_np):
    """200 forwards at B = 64 (counters) and B = 1, 3 (granules), T = 16, while another stream hammers HBM and the CUs: every result must equal the first
    bit for bit (a stale read of a handed-off state, or a hand-off that depends on arrival order, shows as a difference),
    and the first must match the oracle."""
    from oracle import tepose_ref as O
    model, state, _ = _model(2, 1024, 3, smpl_np)
    x = synth.synthetic_windows(B, 16, 5)
    xd = torch.from_numpy(x).cuda()
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device='cuda')
    with torch.no_grad():
        first = model.encoder(xd).clone()
        for it in range(200):
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    junk.mul_(1.0001).add_(0.5)
                    (junk[:1 << 22].view(2048, 2048) @ junk[1 << 22:1 << 23].view(2048, 2048)).sum()
            f = model.encoder(xd)
            assert torch.equal(f, first), it
    torch.cuda.synchronize()
    enc, _ = O.split_state_dict(state, torch.float64)
    with torch.no_grad():
        ref = O.encoder_fwd(enc, torch.from_numpy(x).double(), 2)
    assert (first.cpu().double() - ref).abs().max() < 2e-5


def _run_env(env_extra, code):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


_CODE = """
import numpy as np, torch, sys
from tepose_amd import synth
from tepose_amd.testing import build_model
model, state, smpl_np = build_model(2, 1024, seed=4, device='cuda')
out = {}
for B, T in ((1, 16), (3, 5), (4, 32), (40, 6)):
    x = torch.from_numpy(synth.synthetic_windows(B, T, 9)).cuda()
    with torch.no_grad():
        o = model(x, J_regressor=torch.from_numpy(smpl_np['J_regressor_h36m']))[0]
    out['%d_%d' % (B, T)] = np.concatenate([o['theta'].cpu().numpy().ravel(), o['verts'].cpu().numpy().ravel()[::97]])
np.savez(sys.argv[1], **out)
"""


def test_persistent_and_step_per_launch_paths_agree(tmp_path, smpl_np):
    """Same model, same windows: the library with the persistent kernel (default) and with it disabled
    (TEPOSE_SEQ_MAX_M=0 -> step-per-launch kernels), full forward incl. B <= 4."""
    a, b = str(tmp_path / 'a.npz'), str(tmp_path / 'b.npz')
    _run_env({}, _CODE.replace('sys.argv[1]', repr(a)))
    _run_env({'TEPOSE_SEQ_MAX_M': '0'}, _CODE.replace('sys.argv[1]', repr(b)))
    ga, gb = np.load(a), np.load(b)
    for k in ga.files:
        assert np.abs(ga[k] - gb[k]).max() < 2e-5, k


# ---------------------------------------------------------------- persistent regressor kernel (csrc/reg_seq.hip)
@pytest.mark.parametrize('N,n_iter,use_j', [(1, 3, True), (5, 3, False), (16, 3, True), (17, 2, True), (33, 3, False),
                                             (64, 3, True), (48, 1, True), (7, 0, False)])
def test_regressor_on_persistent_kernel_vs_oracle(N, n_iter, use_j, smpl_np):
    """N <= 64 rows: fc1 / fc2 / decoders x n_iter in one launch (64 workgroups, hand-offs through L2)."""
    from oracle import tepose_ref as O
    model, state, _ = _model(1, 64, 3, smpl_np)
    feat = synth.normal('rsfeat%d' % N, (N, 2048), std=0.5)
    J = smpl_np['J_regressor_h36m'] if use_j else None
    with torch.no_grad():
        out = model.regressor(torch.from_numpy(feat).cuda(), n_iter=n_iter,
                              J_regressor=None if J is None else torch.from_numpy(J))[0]
    _, reg = O.split_state_dict(state, torch.float64)
    smpl = O.smpl_tensors(smpl_np, torch.float64)
    with torch.no_grad():
        ref = O.regressor_fwd(reg, smpl, torch.from_numpy(feat).double(), None if J is None else torch.from_numpy(J).double(),
                              n_iter=n_iter)
    for k in ('rotmat', 'verts', 'kp_3d', 'kp_2d'):
        assert (out[k].cpu().double() - ref[k]).abs().max() < 1e-4, k
    assert (out['theta'][:, :3].cpu().double() - ref['theta'][:, :3]).abs().max() < 1e-5
    assert (out['theta'][:, 75:].cpu().double() - ref['theta'][:, 75:]).abs().max() < 1e-5


def test_regressor_persistent_kernel_repeats_bit_identically(smpl_np):
    model, _, _ = _model(1, 64, 3, smpl_np)
    feat = torch.from_numpy(synth.normal('rsrep', (40, 2048), std=0.5)).cuda()
    side = torch.cuda.Stream()
    junk = torch.randn(32 << 20, device='cuda')
    with torch.no_grad():
        first = model.regressor(feat)[0]['theta'].clone()
        for it in range(150):
            if it % 4 == 0:
                with torch.cuda.stream(side):
                    junk.mul_(1.0001)
            assert torch.equal(model.regressor(feat)[0]['theta'], first), it
    torch.cuda.synchronize()
