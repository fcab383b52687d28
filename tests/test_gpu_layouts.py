"""GPU: internal data layouts of large batches change WHERE bytes live, never the arithmetic.
  * gate pre-activations frame-major + 16 x 16-blocked and fp32 recurrent state blocked between the barrier-free projection kernel and the
    fused GRU step / first-step kernels (TEPOSE_GI_BLK, common.h gi_blk_offset / st_blk_offset): bit-identical to the row-major layout wherever both
    handles run the same step kernel (TEPOSE_GRU_STATE=fp32, or ragged row tiles); with the default plane-fed step kernel (full tiles, blocked layouts only)
    the previous state carries 22 instead of 24 bits: within 5e-7 on features;
  * the blend-shape product of >= 512 persons on the persistent barrier-free kernel with scaled planes (TEPOSE_BLEND16_MIN_N): another
    association of the K sum and a per-matrix instead of a per-element lo scale -- within 5e-6 of the two-accumulator kernel, both within 1e-4 of the oracle."""
import numpy as np
import pytest
import torch

from tepose_amd import synth
from tepose_amd.testing import build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('L,H,B,T', [(2, 128, 2048, 4),      # B * T = 8192: layer 0 blocked + frame-major too; full row tiles: the plane-fed step kernel by default
                                      (3, 256, 2064, 4),      # ragged last 128-row tile (2064 = 16 * 128 + 16), three layers
                                      (2, 64, 2050, 5),       # B % 16 != 0: layer 0 stays row-major, layers >= 1 and the state are blocked
                                      (2, 192, 700, 3),       # just above the scaled-format threshold (640), Hp = 192
                                      (1, 320, 1000, 2)])     # one layer: only the state and the first steps
@pytest.mark.parametrize('gru_state', ['fp32', 'planes'])
def test_blocked_operands_are_bit_identical_to_row_major(monkeypatch, L, H, B, T, gru_state):
    smpl_np = synth.synthetic_smpl(0)
    state = synth.synthetic_state_dict(L, H, 5)
    monkeypatch.setenv('TEPOSE_GRU_STATE', gru_state)
    monkeypatch.setenv('TEPOSE_GI_BLK', '0')
    plain, _, _ = build_model(L, H, seed=5, device='cuda', smpl_np=smpl_np, state=state)
    monkeypatch.delenv('TEPOSE_GI_BLK')
    blocked, _, _ = build_model(L, H, seed=5, device='cuda', smpl_np=smpl_np, state=state)
    monkeypatch.delenv('TEPOSE_GRU_STATE')
    sp, sb = plain._engine.select_kernels(B, T), blocked._engine.select_kernels(B, T)
    same_kernel = all(sp.get(k) == sb.get(k) for k in ('gru_step', 'gru_step_l1'))
    assert same_kernel == (gru_state == 'fp32' or B % 128 != 0 or L == 1)
    x = torch.from_numpy(synth.synthetic_windows(B, T, 17)).cuda()
    with torch.no_grad():
        fa = plain.encoder(x)
        fb = blocked.encoder(x)
        fa_tr = plain.encoder(x, is_train=True)
        fb_tr = blocked.encoder(x, is_train=True)
    assert torch.isfinite(fb).all()
    if same_kernel:
        assert torch.equal(fa, fb), (L, H, B, T, float((fa - fb).abs().max()))
        assert torch.equal(fa_tr, fb_tr), (L, H, B, T)
    else:
        assert float((fa - fb).abs().max()) < 5e-7 and float((fa_tr - fb_tr).abs().max()) < 5e-7, (L, H, B, T, float((fa - fb).abs().max()))


def test_blend_shape_product_on_the_persistent_kernel(monkeypatch):
    from oracle import tepose_ref as O
    L, H, B, T = 2, 64, 1100, 3
    smpl_np = synth.synthetic_smpl(0)
    state = synth.synthetic_state_dict(L, H, 6)
    J = smpl_np['J_regressor_h36m']
    monkeypatch.setenv('TEPOSE_BLEND16_MIN_N', str(0x7fffffff))
    old, _, _ = build_model(L, H, seed=6, device='cuda', smpl_np=smpl_np, state=state)
    monkeypatch.delenv('TEPOSE_BLEND16_MIN_N')
    new, _, _ = build_model(L, H, seed=6, device='cuda', smpl_np=smpl_np, state=state)
    xn = synth.synthetic_windows(B, T, 23)
    x = torch.from_numpy(xn).cuda()
    with torch.no_grad():
        oa = old(x, J_regressor=torch.from_numpy(J))[0]
        ob = new(x, J_regressor=torch.from_numpy(J))[0]
    for k in ('theta', 'rotmat'):
        assert torch.equal(oa[k], ob[k]), k                      # nothing upstream of the blend shapes changes
    for k in ('verts', 'kp_3d', 'kp_2d'):
        assert float((oa[k] - ob[k]).abs().max()) < 5e-6, (k, float((oa[k] - ob[k]).abs().max()))
    ref = O.tepose_fwd(state, smpl_np, xn[:24], L, J_regressor=J)
    for k in ('verts', 'kp_3d'):
        assert np.abs(ob[k][:24].cpu().numpy() - ref[k].numpy()).max() < 1e-4, k
