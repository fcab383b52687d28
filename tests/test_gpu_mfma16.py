"""GPU: the scaled-plane kernels on the other fp16 MFMA shape (csrc/gemm_h3s16.hip, v_mfma_f32_16x16x32_f16; TEPOSE_MFMA16 bit 1 =
plain products, bit 2 = fused GRU step).  One MFMA spans two K-tiles there, so the sums are associated differently from the
32x32x16 kernels: results agree to rounding, not bit for bit, and both must sit within the oracle's tolerance.  The knob is read
when a handle is created, so both settings run in one process."""
import numpy as np
import pytest
import torch

from oracle import tepose_ref as O
from tepose_amd import synth
from tepose_amd.testing import build_model

pytestmark = pytest.mark.gpu

# (L, H, B, T): row tiles of 128 with ragged last tiles, 1 / 2 / 3 layers (2- and 3-direction step launches), hidden sizes
# whose unit tiles (64) do not fill the 4 x 8 walk, B at / just above the scaled-format threshold (2048)
SHAPES = [(2, 1024, 2305, 3), (1, 320, 2050, 2), (2, 192, 2100, 3), (3, 512, 2049, 2), (2, 256, 4096, 4)]


@pytest.mark.parametrize('knob', ['1', '2', '3', '5', '13', '25'])     # bit 1 plain products, bit 2 GRU step (8 waves), bit 4 GRU step (4 waves of 64 x 96), bit 8 plain products barrier-free, bit 16 GRU step persistent + barrier-free
def test_mfma16_kernels_against_the_default_shape_and_the_fp64_oracle(monkeypatch, knob):
    from tepose_amd import _lib
    errs0 = int(_lib.load().tepose_debug_kernel_errors())            # process-wide counter (tests/test_gpu_status.py forces give-ups on purpose)
    for L, H, B, T in SHAPES:
        smpl_np = synth.synthetic_smpl(0)
        state = synth.synthetic_state_dict(L, H, 11)
        monkeypatch.setenv('TEPOSE_MFMA16', '0')                           # every scaled-plane kernel on 32x32x16
        base, _, _ = build_model(L, H, seed=11, device='cuda', smpl_np=smpl_np, state=state)
        monkeypatch.setenv('TEPOSE_MFMA16', knob)
        alt, _, _ = build_model(L, H, seed=11, device='cuda', smpl_np=smpl_np, state=state)
        monkeypatch.delenv('TEPOSE_MFMA16', raising=False)
        x = torch.from_numpy(synth.synthetic_windows(B, T, 42)).cuda()
        with torch.no_grad():
            fa = base.encoder(x).cpu().numpy()
            fb = alt.encoder(x).cpu().numpy()
            fb2 = alt.encoder(x).cpu().numpy()
        assert np.array_equal(fb, fb2), (L, H, B, T)                      # deterministic
        assert np.abs(fa - fb).max() < 5e-6, (L, H, B, T, np.abs(fa - fb).max())
        enc, _ = O.split_state_dict(state, torch.float64)
        with torch.no_grad():
            ref = O.encoder_fwd(enc, torch.from_numpy(synth.synthetic_windows(B, T, 42)[:48]).double(), L).numpy()
        assert np.abs(fb[:48] - ref).max() < 2e-5, (L, H, B, T)
        rows = np.r_[B - 48:B]                                             # the ragged last row tile too
        with torch.no_grad():
            ref2 = O.encoder_fwd(enc, torch.from_numpy(synth.synthetic_windows(B, T, 42)[rows]).double(), L).numpy()
        assert np.abs(fb[rows] - ref2).max() < 2e-5, (L, H, B, T)
    assert int(_lib.load().tepose_debug_kernel_errors()) == errs0    # no wave of the barrier-free kernels ever gave up a poll
