"""GPU: the multi-rank paths on a 1-GPU box -- two ranks share cuda:0 and talk over gloo (the collectives are the
same calls RCCL serves on a real node).  Both entry points are started the way the driver starts them: a plain
`python <script> --gpus 2`, which must spawn its own ranks (reference shard axis evaluate.py:214)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]            # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_and_they_agree():
    B, steps = 256, 2
    r = _run(['bench.py', '--gpus', '2', '--backend', 'gloo', '--share-device0', '--batch', str(B), '--steps', str(steps),
              '--warmup', '1', '--no-cpu-baseline', '--no-extra'])
    assert r['n_gpus'] == 2 and r['steps'] == steps and r['scaling'] == 'weak'
    assert r['ranks_agree'] is True                      # every rank's probe forward on the broadcast blob
    assert r['weight_broadcast_ms'] > 0 and r['weight_blob_MB'] > 100
    assert r['weight_broadcast_MB'] < 0.45 * r['weight_blob_MB']          # only the fp32 sections travel; planes re-derived per rank
    assert len(r['per_rank']) == 2
    assert sorted(int(x[0]) for x in r['per_rank']) == [0, 1]
    assert sum(int(x[2]) for x in r['per_rank']) == 2 * B * steps          # N * B windows per step counted
    assert all(x[3] == 1.0 for x in r['per_rank'])       # finite outputs on every rank
    assert abs(r['value'] - 2 * B * steps / (r['ms_per_step'] * steps / 1e3)) < 1e-6 * r['value']
    assert r['outputs_finite'] is True
    # all_gather of the device identities: two ranks, ONE physical GPU here (an 8-GPU run must report 8)
    assert r['n_ranks_seen'] == 1 and len(r['rank_devices']) == 2 and r['rank_devices'][0] == r['rank_devices'][1]


def test_single_rank_bench_path_is_unchanged():
    r = _run(['bench.py', '--batch', '64', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extra'])
    assert r['n_gpus'] == 1 and 'weight_broadcast_ms' not in r and r['outputs_finite'] is True


def test_rccl_path_executes_at_world_size_one():
    """`--force-dist`: dist.init_process_group('nccl', device_id=...) -- RCCL -- plus the blob broadcast, the all_reduce of
    the step time and the gather of the per-rank records, on this 1-GPU box: the collectives of the 8-GPU run execute at
    least once before the driver's node does it for real (reference shard axis: evaluate.py:214).  A second handle adopts
    a copy of the broadcast blob and must reproduce rank 0's probe forward bit for bit."""
    r = _run(['bench.py', '--force-dist', '--batch', '128', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extra'])
    assert r['n_gpus'] == 1 and r['dist_backend'] == 'nccl'
    assert r['weight_broadcast_ms'] > 0 and r['weight_blob_MB'] > 100 and r['weight_broadcast_MB'] < 0.45 * r['weight_blob_MB']
    assert r['adopted_blob_matches'] is True and r['ranks_agree'] is True
    assert len(r['per_rank']) == 1 and int(r['per_rank'][0][2]) == 128 * 2 and r['per_rank'][0][3] == 1.0
    assert r['outputs_finite'] is True


def test_clip_sharded_evaluation_runs_on_rccl_at_world_size_one():
    common = ['tools/evaluate_clips.py', '--layers', '1', '--hidden', '64', '--seqlen', '5', '--clips', '7', '--min-len', '6',
              '--max-len', '40']
    one = _run(common)
    rccl = _run(common + ['--force-dist'])
    assert rccl['dist_backend'] == 'nccl' and rccl['n_gpus'] == 1
    assert rccl['metrics_mm'] == one['metrics_mm']         # same clips, same batches: the gathers only move the records
    assert rccl['per_rank']['clips'] == [7]
    # the evaluation tool runs the weight collective too (both models: TePose and the VIBE bootstrap)
    assert rccl['weight_broadcast_ms'] > 0 and 0 < rccl['weight_broadcast_MB'] < rccl['weight_blob_MB'] and rccl['n_ranks_seen'] == 1


def test_clip_sharded_evaluation_world2_equals_world1():
    common = ['tools/evaluate_clips.py', '--layers', '1', '--hidden', '64', '--seqlen', '5', '--clips', '7', '--min-len', '6',
              '--max-len', '40']
    one = _run(common)
    two = _run(common + ['--gpus', '2', '--backend', 'gloo', '--share-device0'])
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2 and one['clips'] == two['clips'] == 7
    for k in ('mpjpe', 'mpjpe_pa', 'accel_err', 'mpvpe'):
        # same clips, each processed whole on one rank: only the lock-step batch composition differs (rounding of
        # different kernels, amplified by the theta feedback of ~35 window steps)
        assert abs(one['metrics_mm'][k] - two['metrics_mm'][k]) < 2e-5 * abs(one['metrics_mm'][k]) + 1e-3, k
    # rank 1 never loaded a weight: it ran on the blobs broadcast by rank 0 (TePose + VIBE), planes re-derived on its side
    assert two['weight_broadcast_ms'] > 0 and two['n_ranks_seen'] == 1 and len(two['rank_devices']) == 2
    st = two['per_rank']
    assert len(st['seconds']) == 2 and sum(st['clips']) == 7 and sum(st['frames']) == two['frames']
    assert st['seconds_max_over_mean'] >= 1.0 and two['imbalance_max_over_mean_rank_frames'] >= 1.0


def test_two_processes_sharing_the_gpu_get_bit_identical_results():
    """Two processes on ONE GPU, each repeating every module of the small evaluation pipeline and comparing bitwise with
    its own first result.  With packed-fp32 VALU instructions in the binary 1-3 % of the SMPL launches came back with a
    wrong x for 16 consecutive vertices (gfx950 under context switching; DESIGN.md "shared-GPU erratum") -- the library is
    built without them (__graft_entry__.HIPCC_EXTRA), and this must stay at zero mismatches."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    procs = [subprocess.Popen([sys.executable, 'tools/race_probe.py', ROOT, '700'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for _ in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        last = [l for l in out.splitlines() if l.startswith('mismatches')][-1]
        assert all(int(v) == 0 for v in last.replace('}', '').split(':')[1:] for v in [v.split(',')[0]]), out[-1500:]


def test_adopt_blob_rejects_a_foreign_blob_and_accepts_its_own():
    """The packed blob carries a header (magic, ABI, model kind, L, H, layout size, packed sections): a blob received by
    broadcast is adopted only by a handle of the same model -- a TePose blob must not configure a VIBE handle, nor a handle
    of another size (tepose_adopt_blob used to set every packed flag blindly)."""
    import torch
    from tepose_amd import _lib, synth
    from tepose_amd.engine import Engine
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(1, 64, seed=2, device='cuda', smpl_np=smpl_np)
    x = torch.from_numpy(synth.synthetic_windows(2, 4, 3)).cuda()
    with torch.no_grad():
        ref = model(x)[0]['verts'].clone()
    blob = model._engine.blob
    lib = _lib.load()
    # same model: a fresh handle adopts the blob and reproduces the forward
    twin, _, _ = build_model(1, 64, seed=99, device='cuda', smpl_np=smpl_np)          # other weights: would differ if it packed its own
    twin._engine.adopt_blob(blob.clone(), twin)
    with torch.no_grad():
        assert torch.equal(twin(x)[0]['verts'], ref)
    # other size / other kind: refused with TEPOSE_E_STATE (-4)
    for eng in (Engine(2, 64), Engine(1, 128), Engine(1, 64, 'vibe')):
        big = torch.zeros(max(eng.packed_bytes, blob.numel()), dtype=torch.uint8, device='cuda')
        big[:blob.numel()] = blob
        assert lib.tepose_set_blob(eng.handle, big.data_ptr(), big.numel()) == 0
        assert lib.tepose_adopt_blob(eng.handle) == -4
    # garbage: refused
    eng = Engine(1, 64)
    junk = torch.randint(0, 255, (eng.packed_bytes,), dtype=torch.uint8, device='cuda')
    assert lib.tepose_set_blob(eng.handle, junk.data_ptr(), junk.numel()) == 0
    assert lib.tepose_adopt_blob(eng.handle) == -4


@pytest.mark.parametrize('L,H', [(2, 128), (1, 64), (2, 1024)])
def test_planes_derived_from_the_fp32_sections_are_bit_identical(L, H):
    """Broadcast less: a receiver that gets only the blob's fp32 ranges (tepose_fp32_ranges: header, packed
    matrices, tables, collapsed maps) and rebuilds the hi / lo planes itself (tepose_derive_planes) must end up with the SAME
    BYTES as the packing rank's blob, and the same forward."""
    import torch
    from tepose_amd import synth
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(L, H, seed=5, device='cuda', smpl_np=smpl_np)
    x = torch.from_numpy(synth.synthetic_windows(70, 4, 3)).cuda()        # split-precision path (planes in use)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        ref = model(x, J_regressor=J)[0]
    src = model._engine.blob
    ranges = model._engine.fp32_ranges()
    sent = sum(n for _, n in ranges)
    assert 0 < sent < (0.4 if H == 1024 else 0.5) * src.numel()             # under half of the blob travels (36 % at the published size)
    twin, _, _ = build_model(L, H, seed=77, device='cuda', smpl_np=smpl_np)  # other weights: must not matter
    blob = torch.zeros_like(src)
    for off, n in ranges:
        blob[off:off + n] = src[off:off + n]
    twin._engine.adopt_blob(blob, twin, derive=True)
    torch.cuda.synchronize()
    assert torch.equal(blob, src)
    with torch.no_grad():
        out = twin(x, J_regressor=J)[0]
    for k in ref:
        assert torch.equal(out[k], ref[k]), k


def test_two_processes_running_persistent_kernels_stay_bit_identical():
    """Both processes run the persistent recurrent / regressor kernels (n_layers = 2, hidden = 1024, B = 1 .. 64) on ONE GPU:
    every forward must equal the process's first one bit for bit -- no hand-off may depend on who else holds CUs, and no
    bounded wait may expire (launches need at most 192 of the 256 CUs by default)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    procs = [subprocess.Popen([sys.executable, 'tools/soak_seq.py', '240'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for _ in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, err[-2000:]
        last = [l for l in out.splitlines() if l.startswith('soak mismatches')][-1]
        counts = [int(v.split(',')[0].strip(' }')) for v in last.split(':')[1:]]
        assert len(counts) >= 10 and sum(counts) == 0, last
