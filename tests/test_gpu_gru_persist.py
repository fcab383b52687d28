"""GPU: the opt-in persistent form of the fused GRU step (csrc/gemm_h3s.hip gru_h3s_persist_kernel, TEPOSE_GRU_PERSIST=1: 256-row
tiles walked by 256 persistent workgroups) must be bit-identical to the default one-workgroup-per-tile form -- same fragments,
same K order, same accumulators -- and therefore as close to the oracle.  The switch is read once per process, so each setting
runs in its own interpreter."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tepose_amd import synth
from tepose_amd.testing import build_model
out = {}
for L, H, B, T in [(2, 1024, 2305, 3), (1, 320, 2050, 2), (2, 192, 2100, 3), (2, 256, 2304, 3), (3, 512, 2049, 2), (2, 1024, 2048, 2)]:
    model, _, _ = build_model(L, H, seed=11, device='cuda', smpl_np=synth.synthetic_smpl(0))
    x = torch.from_numpy(synth.synthetic_windows(B, T, 42)).cuda()
    with torch.no_grad():
        out['f_%%d_%%d_%%d_%%d' %% (L, H, B, T)] = model.encoder(x).cpu().numpy()
np.savez(sys.argv[1], **out)
''' % ROOT


def _run(persist, path):
    env = dict(os.environ, TEPOSE_GRU_PERSIST=str(persist), TEPOSE_MFMA16='1')      # the 32x32x16 step kernels (the persistent form's twin)
    p = subprocess.run([sys.executable, '-c', SCRIPT, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return np.load(path)


def test_persistent_gru_step_is_bit_identical_and_matches_the_oracle(tmp_path):
    import torch
    from oracle import tepose_ref as O
    from tepose_amd import synth
    a = _run(0, str(tmp_path / 'a.npz'))
    b = _run(1, str(tmp_path / 'b.npz'))
    assert sorted(a.files) == sorted(b.files) and len(a.files) == 6
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
        _, L, H, B, T = k.split('_')
        state = synth.synthetic_state_dict(int(L), int(H), 11)
        enc, _ = O.split_state_dict(state, torch.float64)
        x = synth.synthetic_windows(int(B), int(T), 42)
        with torch.no_grad():
            ref = O.encoder_fwd(enc, torch.from_numpy(x[:64]).double(), int(L))
        assert np.abs(b[k][:64] - ref.numpy()).max() < 2e-5, k
