"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Tolerance 1e-4 abs fp32 on theta / verts / joints (BASELINE.json north_star)."""
import glob
import os

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'tepose_*.npz')))
TOL = 1e-4


@pytest.fixture(scope='module')
def smpl_np():
    return synth.synthetic_smpl(0)


def _model(L, H, seed, smpl_np):
    from tepose_amd.testing import build_model
    return build_model(L, H, seed=seed, device='cuda', smpl_np=smpl_np)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---------------------------------------------------------------- GEMM building block
@pytest.mark.parametrize('M,N,K,relu,bias', [
    (128, 128, 32, 0, 0), (1, 5, 64, 0, 1), (130, 200, 96, 1, 1), (777, 3072, 1024, 0, 1),
    (64, 157, 1024, 0, 0), (4096, 256, 2144, 1, 1)])
def test_gemm_f32_matches_fp64(M, N, K, relu, bias):
    from tepose_amd import _lib
    lib = _lib.load()
    A = synth.normal('gA%d' % M, (M, K))
    W = synth.normal('gW%d' % N, (N, K))
    b = synth.normal('gb%d' % N, (N,))
    dA, dW, db = _dev(A), _dev(W), _dev(b)
    C = torch.full((M, N), float('nan'), device='cuda')
    ws = torch.empty(lib.tepose_gemm_workspace_bytes(N, K), dtype=torch.uint8, device='cuda')
    rc = lib.tepose_gemm_f32(dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr() if bias else None, C.data_ptr(), N,
                             M, N, K, relu, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    A64 = np.maximum(A, 0).astype(np.float64) if relu else A.astype(np.float64)
    ref = A64 @ W.astype(np.float64).T + (b.astype(np.float64) if bias else 0)
    err = np.abs(C.cpu().numpy() - ref).max()
    assert err < 2e-6 * K ** 0.5 * 4, err   # fp32 fmaf chain vs fp64


def test_gemm_rejects_bad_arguments():
    from tepose_amd import _lib
    lib = _lib.load()
    t = torch.zeros(64, 33, device='cuda')
    assert lib.tepose_gemm_f32(t.data_ptr(), 33, t.data_ptr(), 33, None, t.data_ptr(), 64, 64, 64, 33, 0,
                               t.data_ptr(), 16, None) == -2


# ---------------------------------------------------------------- modules vs oracle
@pytest.mark.parametrize('L,H,B,T', [(2, 1024, 2, 6), (1, 128, 3, 5), (2, 256, 130, 4), (3, 64, 2, 4),
                                      (1, 100, 2, 3), (2, 1024, 1, 32), (2, 128, 40, 7), (1, 64, 17, 4),
                                      (2, 128, 3, 1), (1, 64, 2, 2), (2, 64, 900, 2),
                                      (1, 2048, 2, 6), (1, 2048, 800, 3), (3, 64, 777, 2),
                                      (2, 64, 100, 1), (2, 100, 120, 3), 
                                      (2, 128, 70, 9), (2, 1024, 40, 16), (2, 64, 333, 4), (2, 192, 129, 4),   # 512 <= B*T < 8192, L >= 2: 128 x 288 tiles (ragged last row tile)
                                      (2, 64, 1100, 8), (3, 100, 1030, 8), (2, 64, 4100, 3), (2, 128, 2048, 2),   # B*T >= 8192: single-accumulator
                                      # layer-0 projection; B >= 640 (TEPOSE_S_MIN_B; 2048 until round 4): scaled-format recurrent path; class defaults: n_layers=1, hidden=2048
                                      # B >= 640: the fused GRU step of large batches (gru_step16_kernel: 16x16x32 MFMA, four waves of 64 x 96; 128-row tiles,
                                      # full -> the plane-fed instantiation, ragged -> the fp32-state one; unit-tile counts 3, 4, 5, 8, 16) and the layer >= 1
                                      # projections on gemm_h3s_persist16c_kernel (barrier-free); first steps on gru_first16_kernel where Hp % 128 == 0
                                      (2, 192, 2100, 3), (1, 320, 2050, 2), (3, 512, 2049, 2), (2, 1024, 2048, 2),
                                      (2, 1024, 2305, 3),
                                      # B * T >= 8192 AND B >= 640 AND B % 16 == 0: layer-0 gate pre-activations frame-major + 16 x 16-blocked (common.h gi_blk_offset),
                                      # layers >= 1 blocked whenever B >= 640; a ragged last 128-row tile (2064 = 16 * 128 + 16), three layers
                                      (2, 128, 2048, 4), (3, 256, 2064, 4), 
                                      # B >= 640 AND B % 128 == 0 (round 5): every row tile full -> gru_step16_kernel<true> (cell operands through the LDS-DMA
                                      # stream, h_{t-1} rebuilt from the state planes): two- and three-direction launches, 2 / 3 / 4 layers, unit-tile counts 1 ... 16,
                                      # T from 2 (one step behind the first) to 7
                                      (2, 256, 1024, 6), (3, 128, 1280, 5), (2, 1024, 640, 4), (2, 64, 768, 7), (4, 64, 640, 3)])
def test_encoder_vs_oracle(L, H, B, T, smpl_np):
    from oracle import tepose_ref as O
    model, state, _ = _model(L, H, 11, smpl_np)
    x = synth.synthetic_windows(B, T, 42)
    with torch.no_grad():
        feat = model.encoder(_dev(x))
        feat_tr = model.encoder(_dev(x), is_train=True)
    enc, _ = O.split_state_dict(state, torch.float64)
    # windows are independent rows: for large batches of wide models the fp64 oracle runs on 40 rows of the first and of the last 128-row tile (the ragged one) and
    # on 40 rows drawn from the rest -- every kernel treats all row tiles alike, and the whole-batch outputs are checked for finiteness
    rows = np.arange(B) if B * H <= 150000 else np.unique(np.r_[0:40, B - 40:B, np.random.RandomState(B).randint(0, B, 40)])
    with torch.no_grad():
        ref = O.encoder_fwd(enc, torch.from_numpy(x[rows]).double(), L)
        ref_tr = O.encoder_fwd(enc, torch.from_numpy(x[rows]).double(), L, is_train=True)
    assert feat.shape == (B, 2048) and feat_tr.shape == (B, 2, 2048)
    assert torch.isfinite(feat).all() and torch.isfinite(feat_tr).all()
    assert (feat.cpu().double()[rows] - ref).abs().max() < 2e-5
    assert (feat_tr.cpu().double()[rows] - ref_tr).abs().max() < 2e-5


@pytest.mark.parametrize('N,use_j', [(1, True), (5, False), (200, True), (1000, True)])   # 1000: split-precision FCs
def test_regressor_vs_oracle(N, use_j, smpl_np):
    from oracle import tepose_ref as O
    model, state, _ = _model(1, 64, 3, smpl_np)
    feat = synth.normal('feat%d' % N, (N, 2048), std=0.5)
    J = smpl_np['J_regressor_h36m'] if use_j else None
    with torch.no_grad():
        out = model.regressor(_dev(feat), J_regressor=None if J is None else torch.from_numpy(J))[0]
    _, reg = O.split_state_dict(state, torch.float64)
    smpl = O.smpl_tensors(smpl_np, torch.float64)
    with torch.no_grad():
        ref = O.regressor_fwd(reg, smpl, torch.from_numpy(feat).double(),
                              None if J is None else torch.from_numpy(J).double())
    for k in ('rotmat', 'verts', 'kp_3d', 'kp_2d'):
        assert out[k].shape == ref[k].shape, k
        assert (out[k].cpu().double() - ref[k]).abs().max() < TOL, k
    _check_theta(out['theta'].cpu().double().numpy(), ref['theta'].numpy(), ref['rotmat'].numpy())


def _check_theta(theta, ref_theta, ref_R):
    """cam and betas directly; axis-angle directly away from the pi singularity, where
    R -> aa is ill-conditioned (SURVEY.md hard part 5) the rotation itself was compared."""
    assert np.abs(theta[:, :3] - ref_theta[:, :3]).max() < TOL
    assert np.abs(theta[:, 75:] - ref_theta[:, 75:]).max() < TOL
    aa, raa = theta[:, 3:75].reshape(-1, 3), ref_theta[:, 3:75].reshape(-1, 3)
    ang = np.linalg.norm(raa, axis=1)
    ok = ang < 3.0
    assert ok.mean() > 0.8
    assert np.abs(aa[ok] - raa[ok]).max() < TOL


@pytest.mark.parametrize('case', CASES)
def test_full_forward_vs_reference_golden(case, smpl_np):
    g = np.load(os.path.join(GOLDEN, case + '.npz'))
    L, H, B, T, use_j, seed_w, seed_x = [int(v) for v in g['meta']]
    model, _, _ = _model(L, H, seed_w, smpl_np)
    x = synth.synthetic_windows(B, T, seed_x)
    J = torch.from_numpy(smpl_np['J_regressor_h36m']) if use_j else None     # CPU tensor, as evaluate.py:109
    with torch.no_grad():
        out = model(_dev(x), J_regressor=J)
    assert isinstance(out, list) and len(out) == 1
    o = {k: v.cpu().double().numpy() for k, v in out[0].items()}
    assert o['theta'].shape == (B, 85) and o['verts'].shape == (B, 6890, 3)
    assert o['rotmat'].shape == (B, 24, 3, 3) and o['kp_3d'].shape[1] == (14 if use_j else 49)
    assert np.abs(o['rotmat'] - g['rotmat']).max() < TOL
    assert np.abs(o['verts'][:, ::53] - g['verts_sub']).max() < TOL
    assert np.abs(o['verts'].sum(1) - g['verts_sum']).max() < 5e-3       # sum of 6890 coordinates, each within ~1e-6
    assert np.abs(o['kp_3d'] - g['kp_3d']).max() < TOL
    assert np.abs(o['kp_2d'] - g['kp_2d']).max() < TOL                   # the north star's 1e-4 (measured ~1e-7)
    _check_theta(o['theta'], g['theta'].astype(np.float64), g['rotmat'])
    for v in out[0].values():
        assert v.is_contiguous() and v.dtype == torch.float32 and v.is_cuda


# ---------------------------------------------------------------- drop-in behaviours
def test_train_mode_two_predictions(smpl_np):
    from oracle import tepose_ref as O
    model, state, _ = _model(1, 128, 5, smpl_np)
    B, T = 3, 5
    x = synth.synthetic_windows(B, T, 8)
    with torch.no_grad():
        out = model(_dev(x), is_train=True)[0]
    assert out['theta'].shape == (B, 2, 85) and out['verts'].shape == (B, 2, 6890, 3)
    assert out['kp_3d'].shape == (B, 2, 49, 3) and out['rotmat'].shape == (B, 2, 24, 3, 3)
    enc, reg = O.split_state_dict(state)
    smpl = O.smpl_tensors(smpl_np)
    with torch.no_grad():
        f = O.encoder_fwd(enc, torch.from_numpy(x), 1, is_train=True).reshape(-1, 2048)
        ref = O.regressor_fwd(reg, smpl, f)
    assert (out['verts'].cpu().reshape(-1, 6890, 3) - ref['verts']).abs().max() < TOL


def test_weights_are_repacked_after_load_and_smpl_swap(smpl_np):
    from oracle import tepose_ref as O
    from tepose_amd.smpl import SMPL
    model, state, _ = _model(1, 64, 1, smpl_np)
    x = synth.synthetic_windows(2, 4, 3)
    with torch.no_grad():
        a = model(_dev(x))[0]['verts'].clone()
    # new weights through load_state_dict (evaluate.py:124)
    state2 = synth.synthetic_state_dict(1, 64, 2)
    sd = model.state_dict()
    for k, v in state2.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        b = model(_dev(x))[0]['verts'].clone()
    ref_b = O.tepose_fwd(state2, smpl_np, x, 1)['verts']
    assert (a - b).abs().max() > 1e-3
    assert (b.cpu() - ref_b).abs().max() < TOL
    # gender swap: model.regressor.smpl = SMPL(...) (evaluate.py:130-135)
    smpl2 = synth.synthetic_smpl(1)
    model.regressor.smpl = SMPL.from_tables(smpl2).cuda()
    with torch.no_grad():
        c = model(_dev(x))[0]['verts']
    ref_c = O.tepose_fwd(state2, smpl2, x, 1)['verts']
    assert (c.cpu() - ref_c).abs().max() < TOL


def test_outputs_are_fresh_and_deterministic(smpl_np):
    model, _, _ = _model(1, 64, 1, smpl_np)
    x = _dev(synth.synthetic_windows(4, 6, 3))
    with torch.no_grad():
        a = model(x)[0]
        b = model(x)[0]
    for k in a:
        assert a[k].data_ptr() != b[k].data_ptr()
        assert torch.equal(a[k], b[k]), k


# ---------------------------------------------------------------- full benchmark size
def test_full_size_batch_matches_oracle_on_sampled_rows(smpl_np):
    """BASELINE.json configs[2] size ([8192,16,2133], L=2, H=1024): rows of the big batch are
    independent windows, so (i) sampled rows must match the CPU oracle to 1e-4 and (ii) the same
    rows pushed through the small-batch kernels must agree with the big-batch kernels."""
    from bench import synthetic_windows_device
    from oracle import tepose_ref as O
    model, state, _ = _model(2, 1024, 0, smpl_np)
    B, T = 8192, 16
    x = synthetic_windows_device(B, T, 99, torch.device('cuda'))
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        big = model(x, J_regressor=J)[0]
    for v in big.values():
        assert torch.isfinite(v).all()
    R = big['rotmat'].reshape(-1, 3, 3)
    eye = torch.eye(3, device='cuda')
    assert (R @ R.transpose(1, 2) - eye).abs().max() < 1e-4          # rot6d output is orthonormal
    assert (torch.linalg.det(R) - 1).abs().max() < 1e-4
    idx = torch.tensor([0, 1, 127, 128, 4095, 4096, 8191, 5000, 333, 7777, 2048, 6001])
    with torch.no_grad():
        small = model(x[idx.cuda()].contiguous(), J_regressor=J)[0]
    for k in big:
        assert (big[k][idx.cuda()] - small[k]).abs().max() < 2e-5, k   # tile / split-K order only
    ref = O.tepose_fwd(state, smpl_np, x[idx.cuda()].cpu().numpy(), 2, J_regressor=smpl_np['J_regressor_h36m'])
    for k in ('verts', 'kp_3d', 'rotmat'):
        assert (big[k][idx.cuda()].cpu() - ref[k]).abs().max() < TOL, k
    _check_theta(big['theta'][idx.cuda()].cpu().double().numpy(), ref['theta'].double().numpy(), None)


@pytest.mark.parametrize('L,H,B,T', [(2, 256, 48, 4),     # 192 real rows: layer-1 projections still width-first; 48 rows: three-tile persistent kernel and GEMM
                                      (2, 256, 49, 4),     # 196 rows: tiles; 49 rows: four tiles
                                      (2, 64, 64, 16),     # 1024 rows: the last size of the one-workgroup-per-row input split
                                      (2, 64, 205, 5),     # 1025 rows: the 8-rows-per-block split
                                      (2, 64, 400, 2),     # collapsed regressor product: 25 row tiles x 10 column blocks = 250 one-tile blocks
                                      (2, 64, 416, 2),     # 26 x 10 = 260 > 256: the 64-row form
                                      (1, 64, 16, 3), (1, 64, 17, 3), (2, 128, 33, 2)])   # 1 | 2 | 3 row tiles exactly in the width-first GEMM
def test_round5_threshold_neighbours_match_the_oracle(L, H, B, T, smpl_np):
    """Both sides of every row-count threshold that round 5 added to the small / mid batch dispatch (profiles/r05_mid_rows_gemm.txt, section 4): the
    full forward against the fp64-accumulating CPU oracle on the first and last rows of the batch, finite everywhere."""
    from oracle import tepose_ref as O
    model, state, _ = _model(L, H, 23, smpl_np)
    x = synth.synthetic_windows(B, T, 61)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        out = model(_dev(x), J_regressor=J)[0]
    for v in out.values():
        assert torch.isfinite(v).all()
    rows = np.unique(np.r_[0:min(20, B), max(B - 20, 0):B])
    ref = O.tepose_fwd(state, smpl_np, x[rows], L, J_regressor=smpl_np['J_regressor_h36m'])
    for k in ('verts', 'kp_3d', 'rotmat', 'kp_2d'):
        assert (out[k][rows].cpu() - ref[k]).abs().max() < TOL, (k, float((out[k][rows].cpu() - ref[k]).abs().max()))
    _check_theta(out['theta'][rows].cpu().double().numpy(), ref['theta'].double().numpy(), None)


def test_row_permutation_equivariance(smpl_np):
    model, _, _ = _model(1, 128, 2, smpl_np)
    x = _dev(synth.synthetic_windows(70, 5, 21))
    perm = torch.randperm(70, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        a = model(x)[0]
        b = model(x[perm].contiguous())[0]
    for k in a:
        assert (a[k][perm] - b[k]).abs().max() < 1e-6, k


def test_empty_batch_is_rejected(smpl_np):
    model, _, _ = _model(1, 64, 1, smpl_np)
    with pytest.raises(ValueError):
        model(torch.zeros(0, 6, 2133, device='cuda'))


def test_input_variants_noncontiguous_half_and_jreg_device(smpl_np):
    """Callers hand over slice-assigned / float16-round-tripped tensors (lib/dataset/threedpw_test.py:95-96,137)
    and a CPU J_regressor (evaluate.py:109); all must give the same result as the plain call."""
    model, _, _ = _model(1, 128, 6, smpl_np)
    x = _dev(synth.synthetic_windows(5, 6, 50))
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        base = model(x, J_regressor=J)[0]
        big = torch.zeros(5, 12, 2133, device='cuda')
        big[:, ::2] = x
        nc = model(big[:, ::2], J_regressor=J.cuda())[0]            # strided view + GPU regressor
        xh = x.half()
        h = model(xh, J_regressor=J)[0]                             # fp16 storage -> .float() inside
        href = model(xh.float(), J_regressor=J)[0]
    for k in base:
        assert torch.equal(base[k], nc[k]), k
        assert torch.equal(h[k], href[k]), k


def test_two_models_in_one_process_do_not_share_state(smpl_np):
    from oracle import tepose_ref as O
    a, sa, _ = _model(1, 64, 21, smpl_np)
    b, sb, _ = _model(2, 128, 22, smpl_np)
    x = synth.synthetic_windows(3, 4, 60)
    with torch.no_grad():
        oa1 = a(_dev(x))[0]['verts'].clone()
        ob = b(_dev(x))[0]['verts'].clone()
        oa2 = a(_dev(x))[0]['verts']
    assert torch.equal(oa1, oa2)
    assert (ob.cpu() - O.tepose_fwd(sb, smpl_np, x, 2)['verts']).abs().max() < TOL
    assert (oa1.cpu() - O.tepose_fwd(sa, smpl_np, x, 1)['verts']).abs().max() < TOL


def test_dense_skin_weight_fallback(smpl_np):
    """Tables with more than 4 non-zero skin weights per vertex take the dense skinning kernel;
    the official model's <= 4 take the compacted one.  Both against the oracle."""
    from oracle import tepose_ref as O
    from tepose_amd.testing import build_model
    wide = synth.synthetic_smpl(0, skin_nnz=7)
    assert int((wide['lbs_weights'] != 0).sum(1).max()) > 4 and int((smpl_np['lbs_weights'] != 0).sum(1).max()) <= 4
    x = synth.synthetic_windows(3, 4, 71)
    for tables in (wide, smpl_np):
        model, state, _ = build_model(1, 64, seed=2, device='cuda', smpl_np=tables)
        with torch.no_grad():
            out = model(_dev(x))[0]
        ref = O.tepose_fwd(state, tables, x, 1)
        assert (out['verts'].cpu() - ref['verts']).abs().max() < TOL
        assert (out['kp_3d'].cpu() - ref['kp_3d']).abs().max() < TOL
