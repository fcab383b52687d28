"""GPU: the kernel families a handle's named knobs select for large batches (csrc/api.hip select_kernels) against each other and the fp64 oracle.
  TEPOSE_LARGE_BATCH_KERNELS = scaled (default: gemm_h3s_persist16c_kernel projections, gru_step16_kernel steps) | twoacc (gemm_h3_kernel family everywhere)
  TEPOSE_GRU_STATE           = planes (default: gru_step16_kernel<true> -- cell operands through the LDS-DMA stream, h_{t-1} rebuilt from the state
                               planes, full tiles only) | fp32 (gru_step16_kernel<false>)
Different associations of the K sums (and, for `planes`, a 22-bit previous state): results agree to rounding, each within 2e-5 of the oracle on features;
the give-up counter of the barrier-free kernels stays where it was."""
import numpy as np
import pytest
import torch

from oracle import tepose_ref as O
from tepose_amd import synth
from tepose_amd.testing import build_model

pytestmark = pytest.mark.gpu

# (L, H, B, T): row tiles of 128 full (B % 128 == 0: the `planes` step kernel) and ragged (the fp32-state kernel at every knob), 1 / 2 / 3 layers (2- and
# 3-direction step launches), hidden sizes whose unit tiles (64) do not fill the 4 x 8 walk, B at / just above the scaled-format threshold, long windows
SHAPES = [(2, 1024, 2304, 3), (1, 320, 2050, 2), (2, 192, 2100, 3), (3, 512, 2048, 2), (2, 256, 4096, 4), (2, 128, 640, 16)]


@pytest.mark.parametrize('knobs', [{'TEPOSE_GRU_STATE': 'fp32'}, {'TEPOSE_LARGE_BATCH_KERNELS': 'twoacc'}])
def test_kernel_families_against_the_default_and_the_fp64_oracle(monkeypatch, knobs):
    from tepose_amd import _lib
    errs0 = int(_lib.load().tepose_debug_kernel_errors())            # process-wide counter (tests/test_gpu_status.py forces give-ups on purpose)
    # (the two-accumulator family is the mid-batch default and has its own tests there: four of the six shapes are enough to pin it at large batches)
    for L, H, B, T in (SHAPES if 'TEPOSE_GRU_STATE' in knobs else SHAPES[:3] + SHAPES[5:]):
        smpl_np = synth.synthetic_smpl(0)
        state = synth.synthetic_state_dict(L, H, 11)
        base, _, _ = build_model(L, H, seed=11, device='cuda', smpl_np=smpl_np, state=state)
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        alt, _, _ = build_model(L, H, seed=11, device='cuda', smpl_np=smpl_np, state=state)
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        sel_b, sel_a = base._engine.select_kernels(B, T), alt._engine.select_kernels(B, T)
        if 'TEPOSE_GRU_STATE' in knobs:
            assert sel_a['gru_step'] == 'gru_step16_kernel<false>' and sel_a.get('gru_step_l1', 'gru_step16_kernel<false>') == 'gru_step16_kernel<false>'
            if L >= 2:        # layers >= 1: blocked gate pre-activations from 640 windows on -> plane-fed wherever every row tile is full
                assert sel_b['gru_step_l1'] == ('gru_step16_kernel<true>' if B % 128 == 0 else 'gru_step16_kernel<false>')
            # layer 0: plane-fed only where its projection wrote frame-major blocked gate pre-activations (L >= 2, B * T >= 8192, B % 16 == 0)
            assert sel_b['gru_step'] == ('gru_step16_kernel<true>' if (B % 128 == 0 and L >= 2 and B * T >= 8192) else 'gru_step16_kernel<false>')
        else:
            assert 'h3s' not in sel_a['projection'] and 'step16' not in sel_a['gru_step'] and sel_a['gi1_layout' if L > 1 else 'gi0_layout'] == 'row_major'
        x = torch.from_numpy(synth.synthetic_windows(B, T, 42)).cuda()
        with torch.no_grad():
            fa = base.encoder(x).cpu().numpy()
            fa2 = base.encoder(x).cpu().numpy()
            fb = alt.encoder(x).cpu().numpy()
        assert np.array_equal(fa, fa2), (L, H, B, T)                      # deterministic
        assert np.abs(fa - fb).max() < 5e-6, (L, H, B, T, np.abs(fa - fb).max())
        if 'TEPOSE_GRU_STATE' in knobs and (B % 128 != 0 or L == 1):
            assert np.array_equal(fa, fb)                                 # ragged row tiles / a one-layer model: both handles run the fp32-state kernel
        enc, _ = O.split_state_dict(state, torch.float64)
        rows = np.r_[0:12, B - 12:B]                                       # rows of the first and of the (possibly ragged) last row tile
        with torch.no_grad():
            ref = O.encoder_fwd(enc, torch.from_numpy(synth.synthetic_windows(B, T, 42)[rows]).double(), L).numpy()
        assert np.abs(fa[rows] - ref).max() < 2e-5, (L, H, B, T)
        assert np.abs(fb[rows] - ref).max() < 2e-5, (L, H, B, T)
    assert int(_lib.load().tepose_debug_kernel_errors()) == errs0    # no wave of the barrier-free kernels ever gave up a poll
