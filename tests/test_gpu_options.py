"""GPU: per-handle options (include/tepose_amd.h tepose_set_option; csrc/common.h Options) on live handles -- two models of ONE process with different
thresholds run different kernel families and agree to rounding; an option is refused once the handle has packed anything; the layer-1 projection
profile hook and the joints-from-vertices entry point behave."""
import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu


def test_two_handles_with_different_thresholds_in_one_process():
    from tepose_amd import _lib
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    state = synth.synthetic_state_dict(2, 256, 5)
    a, _, _ = build_model(2, 256, seed=5, device='cuda', smpl_np=smpl_np, state=state)
    b, _, _ = build_model(2, 256, seed=5, device='cuda', smpl_np=smpl_np, state=state)
    b._engine.set_option('S_MIN_B', 4096)                      # b keeps the two-accumulator recurrent path where a switches to the scaled planes
    b._engine.set_option('SEQ_MAX_M', 0)                       # ... and never runs the persistent small-batch kernels
    b._engine.set_option('SMPL_SMALL_MAX_N', 0)
    B, T = 768, 4
    sa, sb = a._engine.select_kernels(B, T), b._engine.select_kernels(B, T)
    assert sa['gru_step_l1'].startswith('gru_step16') and sb['gru_step'] == 'gemm_h3_kernel<GRU>'
    assert a._engine.select_kernels(8, T)['gru_step'].startswith('gru_seq') and b._engine.select_kernels(8, T)['gru_step'] == 'skinny_gru_h3_kernel'
    assert a._engine.select_kernels(2, T)['smpl'] == 'smpl_small_kernel' and b._engine.select_kernels(2, T)['smpl'] != 'smpl_small_kernel'
    for Bq in (768, 8, 2):
        x = torch.from_numpy(synth.synthetic_windows(Bq, T, 9)).cuda()
        with torch.no_grad():
            oa, ob = a(x)[0], b(x)[0]
        for k in ('theta', 'verts', 'kp_3d'):
            assert torch.isfinite(ob[k]).all() and float((oa[k] - ob[k]).abs().max()) < 2e-5, (Bq, k)
    # packed now: the options are frozen (workspace sizes and planes depend on them)
    rc = b._engine.lib.tepose_set_option(b._engine.handle, b'S_MIN_B', 640)
    assert rc == -4                                             # TEPOSE_E_STATE
    with pytest.raises(_lib.TeposeError):
        b._engine.set_option('S_MIN_B', 640)
    assert b._engine.get_option('S_MIN_B') == 4096 and a._engine.get_option('S_MIN_B') == 640


def test_layer1_projection_profile_and_joints_from_vertices():
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(2, 256, seed=6, device='cuda', smpl_np=smpl_np)
    eng = model._engine
    x = torch.from_numpy(synth.synthetic_windows(1024, 4, 10)).cuda()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        out = model(x, J_regressor=J)[0]
        eng.profile_enable(True)
        for _ in range(3):
            model(x, J_regressor=J)
        torch.cuda.synchronize()
    p_ms, p_n, p_fl = eng.profile_read_l1proj()                 # between the two layers' step sequences; read BEFORE the GRU intervals (they reset)
    g_ms, g_n, g_fl = eng.profile_read_gru()
    eng.profile_enable(False)
    assert p_n == 3 and g_n == 3 and 0 < p_ms < g_ms * 10
    H, Bq, T = 256, 1024, 4
    assert abs(p_fl - 2.0 * 3 * H * (Bq * T * H + Bq * T * 2 * H + Bq * 2 * H)) < 1.0
    # evaluate.py:289-291: the H36M regressor's 14 joints of given vertices = what the forward itself regressed from them
    kp = eng.joints_from_verts(out['verts'].contiguous(), J)
    assert kp.shape == (1024, 14, 3) and torch.equal(kp, out['kp_3d'])
    ref = torch.einsum('jv,nvc->njc', J.double(), out['verts'][:64].cpu().double())[:, [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]]
    assert float((kp[:64].cpu().double() - ref).abs().max()) < 1e-5
