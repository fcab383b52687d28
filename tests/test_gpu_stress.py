"""GPU stress: rare-race screen for the LDS-DMA / barrier pipeline of the GEMM and GRU kernels.
A staging race shows as an occasional wrong tile that depends on shape and timing, so: many
shapes against an fp64 product, and bitwise repeatability of identical launches under load."""
import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu


def _gemm(lib, A, W, bias, relu):
    M, K = A.shape
    N = W.shape[0]
    C = torch.full((M, N), float('nan'), device='cuda')
    ws = torch.empty(lib.tepose_gemm_workspace_bytes(N, K), dtype=torch.uint8, device='cuda')
    rc = lib.tepose_gemm_f32(A.data_ptr(), A.stride(0), W.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                             C.data_ptr(), N, M, N, K, relu, ws.data_ptr(), ws.numel(),
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return C


def test_random_shapes_against_fp64():
    from tepose_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(5)
    g = torch.Generator(device='cuda').manual_seed(3)
    for it in range(60):
        M = int(rng.choice([1, 7, 16, 17, 63, 64, 65, 129, 500, 769, 1000, 2049, 5000]))
        N = int(rng.choice([1, 48, 49, 127, 128, 129, 160, 1000, 3072]))
        K = 32 * int(rng.randint(1, 40))
        lda = K + 4 * int(rng.randint(0, 3))                      # padded row stride (16-byte aligned rows)
        Abuf = torch.randn(M, lda, device='cuda', generator=g)
        A = Abuf[:, :K]
        W = torch.randn(N, K, device='cuda', generator=g) * 0.1
        b = torch.randn(N, device='cuda', generator=g)
        relu = int(rng.randint(0, 2))
        C = _gemm(lib, A, W, b, relu)
        ref = (A.double().clamp_min(0) if relu else A.double()) @ W.double().t() + b.double()
        err = (C.double() - ref).abs().max().item()
        assert err < 1e-4 * max(1.0, K ** 0.5 / 8), (it, M, N, K, err)
        assert torch.isfinite(C).all()


def test_big_kernels_are_bitwise_repeatable_under_load():
    """Same launch 12 times, interleaved with other work on the device: any DMA-vs-read race or
    uninitialised read would flip bits somewhere in 0.6 G outputs."""
    from tepose_amd import _lib
    from tepose_amd.testing import build_model
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(9)
    A = torch.randn(16384, 2144, device='cuda', generator=g)
    W = torch.randn(3072, 2144, device='cuda', generator=g) * 0.05
    first = _gemm(lib, A, W, None, 0)
    noise = torch.randn(4096, 4096, device='cuda', generator=g)
    for _ in range(12):
        noise = noise * 1.0001 + 1.0                              # unrelated kernels in between
        again = _gemm(lib, A, W, None, 0)
        assert torch.equal(first, again)
    model, _, _ = build_model(2, 256, seed=3, device='cuda', smpl_np=synth.synthetic_smpl(0))
    x = torch.from_numpy(synth.synthetic_windows(1500, 8, 4)).cuda()       # big-kernel GRU path (M > 768)
    with torch.no_grad():
        ref = model.encoder(x)
        for _ in range(8):
            noise = noise * 0.9999 - 1.0
            assert torch.equal(ref, model.encoder(x))


def test_split_precision_gemm_random_shapes_against_fp64():
    """csrc/gemm_h3.hip: fp16 hi/lo planes, three fp16 MFMAs per k-step, fp32 accumulate.  Same error
    bound as the exact-fp32 kernel on GRU-like operand ranges, incl. subnormal low halves."""
    from tepose_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(11)
    g = torch.Generator(device='cuda').manual_seed(4)
    for it in range(40):
        M = int(rng.choice([1, 63, 255, 256, 257, 769, 1000, 3000]))
        N = int(rng.choice([1, 100, 128, 129, 257, 1000, 3072]))
        K = 32 * int(rng.randint(1, 70))
        scale = float(rng.choice([1e-4, 0.03, 1.0, 30.0]))
        A = torch.randn(M, K, device='cuda', generator=g) * float(rng.choice([0.01, 1.0, 8.0]))
        W = torch.randn(N, K, device='cuda', generator=g) * scale
        b = torch.randn(N, device='cuda', generator=g)
        C = torch.full((M, N), float('nan'), device='cuda')
        ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
        rc = lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, M, N, K,
                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ref = A.double() @ W.double().t() + b.double()
        mag = (A.double().abs() @ W.double().abs().t()).max().item() + 1.0
        err = (C.double() - ref).abs().max().item()
        assert err < 3e-6 * mag, (it, M, N, K, scale, err, mag)      # ~2^-19 of the absolute dot product


def test_split_and_exact_paths_agree_on_a_large_batch(monkeypatch):
    """B > 4 runs the matmuls on the split-precision kernels; TEPOSE_EXACT_FP32=1 (read when the
    handle is created) keeps them on the exact-fp32 MFMA.  Same inputs, both paths: outputs within 1e-5."""
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    x = torch.from_numpy(synth.synthetic_windows(900, 6, 17)).cuda()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    fast, _, _ = build_model(2, 256, seed=5, device='cuda', smpl_np=smpl_np)
    monkeypatch.setenv('TEPOSE_EXACT_FP32', '1')
    exact, _, _ = build_model(2, 256, seed=5, device='cuda', smpl_np=smpl_np)
    monkeypatch.delenv('TEPOSE_EXACT_FP32')
    with torch.no_grad():
        a = fast(x, J_regressor=J)[0]
        b = exact(x, J_regressor=J)[0]
    for k in ('verts', 'kp_3d', 'rotmat', 'kp_2d'):
        d = (a[k] - b[k]).abs().max().item()
        assert 0 < d < 1e-5 or (k != 'verts' and d < 1e-5), (k, d)
    assert (a['theta'][:, :3] - b['theta'][:, :3]).abs().max() < 1e-5


def test_both_numerics_modes_vs_fp64_oracle_at_the_published_size(monkeypatch, capsys):
    """L=2, H=1024, T=16 (the benchmark architecture), B=1024: the split-precision default and the exact-fp32 mode
    against the fp64 oracle on the same windows.  Both must sit far inside the 1e-4 budget; the measured errors are
    printed (pytest -s) -- the split path is not the less accurate of the two."""
    from oracle import tepose_ref as O
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    x = synth.synthetic_windows(1024, 16, 4242)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    errs = {}
    state = None
    for mode in ('split', 'exact'):
        if mode == 'exact':
            monkeypatch.setenv('TEPOSE_EXACT_FP32', '1')
        model, state, _ = build_model(2, 1024, seed=0, device='cuda', smpl_np=smpl_np)
        if mode == 'exact':
            monkeypatch.delenv('TEPOSE_EXACT_FP32')
        with torch.no_grad():
            out = model(torch.from_numpy(x).cuda(), J_regressor=J)[0]
        errs[mode] = {k: out[k][:24].cpu().double() for k in ('verts', 'kp_3d', 'rotmat', 'kp_2d', 'theta')}
        del model
    ref = O.tepose_fwd(state, smpl_np, x[:24], 2, J_regressor=smpl_np['J_regressor_h36m'], dtype=torch.float64)
    for mode, got in errs.items():
        for k in ('verts', 'kp_3d', 'rotmat', 'kp_2d'):
            e = (got[k] - ref[k]).abs().max().item()
            with capsys.disabled():
                print('  %-5s %-7s max|gpu - fp64 oracle| = %.2e' % (mode, k, e))
            assert e < 2e-5, (mode, k, e)
        assert (got['theta'][:, :3] - ref['theta'][:, :3]).abs().max() < 2e-5
        assert (got['theta'][:, 75:] - ref['theta'][:, 75:]).abs().max() < 2e-5


@pytest.mark.parametrize('h3s', ['1'])   # the barrier-free persistent kernel on v_mfma_f32_16x16x32_f16 (gemm_h3s16c.hip): every plain scaled-plane product of large batches
def test_single_accumulator_gemm_persistent_tiles_against_fp64(monkeypatch, h3s):
    """csrc/gemm_h3s16c.hip, the plain scaled-plane product as a persistent kernel (256 workgroups walking the 256 x 256 tiles, pairs of
    K-tiles running on across tile boundaries): more tiles than workgroups, partial edge tiles, unaligned C rows (scalar stores), short K,
    with and without bias."""
    from tepose_amd import _lib
    lib = _lib.load()
    monkeypatch.setenv('TEPOSE_H3S', h3s)
    g = torch.Generator(device='cuda').manual_seed(9)
    cases = [(8192, 2304, 2144 // 32 * 32, 2304, True),      # 32 x 9 = 288 full tiles: 32 workgroups take a second tile
             (5000, 5000, 256, 5000, True),                   # 400 tiles, partial last row / column of tiles
             (4096, 4608, 64, 4608, False),                   # KT = 4: the no-overlap path, 288 tiles
             (2100, 2050, 512, 2051, True),                   # C rows not 16-byte aligned: scalar stores
             (300, 200, 128, 200, False), (256, 256, 1024, 256, True), (65536, 512, 96, 512, True),
             (1500, 1024, 192, 1024, True), (777, 333, 352, 340, True)]  # 6 / 11 pairs of K-tiles: the 16x16x32 walk's short cases
    for M, N, K, ldc, use_bias in cases:
        A = torch.randn(M, K, device='cuda', generator=g) * 3.0             # |a| * 256 < 65504
        W = torch.randn(N, K, device='cuda', generator=g) * 0.05            # |w| * 16384 < 65504
        b = torch.randn(N, device='cuda', generator=g) if use_bias else None
        C = torch.full((M, ldc), float('nan'), device='cuda')
        ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
        rc = lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr() if use_bias else None, C.data_ptr(), ldc,
                                    M, N, K, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ref = A.double() @ W.double().t()
        if use_bias:
            ref += b.double()
        mag = (A.double().abs() @ W.double().abs().t()).max().item() + 1.0
        err = (C[:, :N].double() - ref).abs().max().item()
        assert err < 3e-6 * mag, (M, N, K, err, mag)
        assert torch.isnan(C[:, N:]).all()                                  # nothing written past column N
        C2 = torch.full((M, ldc), float('nan'), device='cuda')
        lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr() if use_bias else None, C2.data_ptr(), ldc,
                               M, N, K, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert torch.equal(C[:, :N], C2[:, :N])                             # deterministic


@pytest.mark.parametrize('h3s', ['1'])
def test_persistent_split_gemm_is_bitwise_repeatable_next_to_a_memory_and_mfma_heavy_stream(monkeypatch, h3s):
    """The persistent plain product (csrc/gemm_h3s16c.hip, gemm_h3s_persist16c_kernel) lets a finished tile's stores stay
    in flight under the next tile's first K-tiles: `s_waitcnt vmcnt(2 Q + 32)` is only correct if memory instructions retire
    from the counter in ISSUE order.  That is the documented behaviour of this part (MI355X_MICROARCH.md, "s_waitcnt": "Loads,
    stores, atomics and LDS-DMA count together, in issue order (flat_* excepted ...)"; the kernel issues global_load_lds and
    global_store only -- checked in the ISA), and this test is the empirical side: a violation would show as a K-tile
    computed on a stage that has not landed, timing-dependent.  So: the kernel 200 times on shapes whose tile walk mixes full tiles (stores overlapped)
    with edge tiles (drained path) and leaves ntiles % 256 != 0, while a second stream keeps HBM (1 GB copies) and the matrix
    pipes (a bf16 GEMM) busy; every result must equal the first bit for bit, and the first must match fp64."""
    from tepose_amd import _lib
    lib = _lib.load()
    monkeypatch.setenv('TEPOSE_H3S', h3s)
    g = torch.Generator(device='cuda').manual_seed(21)
    side = torch.cuda.Stream()
    big_a = torch.empty(256 << 20, dtype=torch.float32, device='cuda')        # 1 GB
    big_b = torch.empty_like(big_a)
    ma = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
    mb = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
    # (M, N, K): 41 x 13 = 533 tiles, last row and column of tiles partial, 2-3 tiles per workgroup in full / edge order
    # mixes; 34 x 9 = 306 full tiles (50 workgroups take a second one); 300 full tiles with a short K (8 K-tiles: the
    # smallest overlapped walk)
    cases = [(10340, 3132, 1024, 140), (8704, 2304, 2144 // 32 * 32, 60), (6400, 3072, 128, 200)]
    for M, N, K, reps in cases:
        A = torch.randn(M, K, device='cuda', generator=g) * 3.0
        W = torch.randn(N, K, device='cuda', generator=g) * 0.05
        b = torch.randn(N, device='cuda', generator=g)
        ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
        st = torch.cuda.current_stream().cuda_stream

        def run():
            C = torch.full((M, N), float('nan'), device='cuda')
            assert lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, M, N, K,
                                          ws.data_ptr(), ws.numel(), st) == 0
            return C
        first = run()
        ref = A.double() @ W.double().t() + b.double()
        mag = (A.double().abs() @ W.double().abs().t()).max().item() + 1.0
        assert (first.double() - ref).abs().max().item() < 3e-6 * mag
        del ref
        bad = 0
        for i in range(reps):
            with torch.cuda.stream(side):                    # competing traffic, re-issued so that it overlaps every repeat
                big_b.copy_(big_a, non_blocking=True)
                torch.matmul(ma, mb)
                big_a.copy_(big_b, non_blocking=True)
            bad += int(not torch.equal(first, run()))
        torch.cuda.synchronize()
        assert bad == 0, (M, N, K, bad)
