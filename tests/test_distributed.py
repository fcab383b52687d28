"""CPU, world_size 2 on gloo: the clip partitioner, the blob broadcast and the padded
per-clip record gather used by the multi-GPU path (one process per GPU on the real node)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tepose_amd.distributed import (StepCost, broadcast_blob, gather_records, imbalance, lockstep_seconds, partition_clips,
                                    predicted_scaling)


def test_partition_is_a_balanced_exact_cover():
    lengths = [1824, 910, 777, 1500, 64, 333, 2048, 12, 905, 640, 1200, 87]
    for world in (1, 2, 4, 8):
        parts = partition_clips(lengths, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(lengths)))          # every clip exactly once, never split
        assert imbalance(lengths, parts) < 1.6
    assert partition_clips(lengths, 8) == partition_clips(list(lengths), 8)    # deterministic
    assert partition_clips([], 4) == [[], [], [], []]


def test_lockstep_cost_model_and_the_partition_built_on_it():
    """The evaluation executor (driver.run_clips) advances a rank's clips in LOCK-STEP (the reference's serial window loop per clip,
    evaluate.py:247-269, run for all clips at once): a rank's time is sum_j step_ms(active clips at step j) -- set by its longest clip --
    not its frame total.  The cost model, the partition that minimises its makespan, and the bound it implies for clip-sharding."""
    from tepose_amd import synth
    sc = StepCost({1: 0.1, 10: 0.15, 40: 0.3})
    assert sc(1) == 0.1 and abs(sc(10) - 0.15) < 1e-12 and abs(sc(25) - 0.225) < 1e-12 and sc(80) > sc(40) and sc(0) == 0.0
    T = 6
    # one clip of 105 frames = 100 windows at step_ms(1); two equal clips: the same 100 steps at step_ms(2)
    assert abs(lockstep_seconds([105], T, sc) - 100 * sc(1) / 1e3) < 1e-12
    assert abs(lockstep_seconds([105, 105], T, sc) - 100 * sc(2) / 1e3) < 1e-12
    # ragged: 10 steps with 3 active, 20 more with 2, 70 more with 1; clips shorter than the window contribute nothing
    assert abs(lockstep_seconds([105, 35, 15, 3], T, sc) - (10 * sc(3) + 20 * sc(2) + 70 * sc(1)) / 1e3) < 1e-12
    assert lockstep_seconds([], T, sc) == 0.0 and lockstep_seconds([4], T, sc) == 0.0
    for seed, n in (('a', 37), ('b', 24), ('c', 9), ('d', 3)):
        lens = [int(v) for v in (300 + 1500 * synth.uniform01('part/' + seed, n))]        # 3DPW-test-like clip lengths
        full = StepCost()
        for world in (1, 2, 4, 8):
            new, old = partition_clips(lens, world, full, T), partition_clips(lens, world)
            assert sorted(i for p in new for i in p) == list(range(n))                    # exact cover, empty ranks allowed
            assert len(new) == world
            mk = lambda parts: max(lockstep_seconds([lens[i] for i in p], T, full) for p in parts)
            assert mk(new) <= mk(old) + 1e-12                                             # never worse than frame-LPT under the model
            assert mk(new) >= (max(lens) - T + 1) * full(1) / 1e3 - 1e-12                 # never below the critical path
        assert partition_clips(lens, 4, full, T) == partition_clips(list(lens), 4, full, T)   # deterministic: every rank computes it alone
    # what the model says about BASELINE config 4 (3DPW-test-like: 37 clips, longest ~1800 frames): the longest clip's serial chain bounds the
    # speed-up of clip sharding far below the GPU count -- only the synthetic weak-scaling line (independent windows) can show >= 6x
    lens = [int(v) for v in (300 + 1500 * synth.uniform01('evalclips', 37))]
    ps = predicted_scaling(lens, T, StepCost())
    assert ps['predicted_seconds']['8'] >= ps['critical_path_seconds'] - 1e-12
    assert 1.0 < ps['predicted_speedup']['8'] < 2.5
    assert ps['predicted_seconds']['8'] <= ps['predicted_seconds_frame_lpt']['8'] + 1e-12
    assert 'NOT a measurement' in ps['model']
    assert partition_clips([], 4, StepCost(), T) == [[], [], [], []]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        blob = (torch.arange(4096, dtype=torch.int64) % 251).to(torch.uint8) if rank == 0 \
            else torch.zeros(4096, dtype=torch.uint8)
        broadcast_blob(blob, src=0)
        assert int(blob.to(torch.int64).sum()) == int((torch.arange(4096) % 251).sum())
        parts = partition_clips(lengths, world)
        mine = parts[rank]
        rec = torch.tensor([[float(i), float(lengths[i]), 0.5 * lengths[i]] for i in mine], dtype=torch.float64)
        rec = rec.reshape(-1, 3)
        out = gather_records(rec, dst=0)
        if rank == 0:
            assert out.shape == (len(lengths), 3)
            ids = sorted(int(v) for v in out[:, 0].tolist())
            assert ids == list(range(len(lengths)))
            # frame-weighted mean formed on rank 0 exactly like evaluate.py:461
            assert abs(float(out[:, 2].sum() / out[:, 1].sum()) - 0.5) < 1e-12
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_broadcast_and_gather_world2_gloo():
    lengths = [300, 20, 75, 1200, 64, 333, 48]     # 7 clips -> ranks hold 3 and 4 (ragged gather)
    mp.spawn(_worker, args=(2, _free_port(), lengths), nprocs=2, join=True)


def _worker_one(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        rec = torch.tensor([[0., 10., 5.], [1., 30., 15.]], dtype=torch.float64)
        out = gather_records(rec, dst=0)              # a group of one rank still goes through all_gather + gather
        assert torch.equal(out, rec)
        out = gather_records(rec[:0], dst=0)          # ... also with no local clip at all
        assert out.shape == (0, 3)
    finally:
        dist.destroy_process_group()


def test_gather_runs_the_collectives_in_a_group_of_one():
    """`--force-dist` (bench.py, tools/evaluate_clips.py): the N-GPU code path at world size 1."""
    mp.spawn(_worker_one, args=(1, _free_port()), nprocs=1, join=True)


def _worker_many(rank, world, port, lengths):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        parts = partition_clips(lengths, world)
        assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
        mine = parts[rank]
        # per-clip records as tepose_amd.evaluate builds them: clip id, frames, sum of per-frame errors, two more sums
        rec = torch.tensor([[float(i), float(lengths[i]), 0.5 * lengths[i], 0.25 * lengths[i], float(rank)] for i in mine],
                           dtype=torch.float64).reshape(-1, 5)
        out = gather_records(rec, dst=0)
        # the step-time reduction of bench.py / evaluate_clips: max over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == float(world)
        if rank == 0:
            assert out.shape == (len(lengths), 5)
            assert sorted(int(v) for v in out[:, 0].tolist()) == list(range(len(lengths)))
            if lengths:
                assert abs(float(out[:, 2].sum() / out[:, 1].sum()) - 0.5) < 1e-12      # frame-weighted mean, evaluate.py:461
            # every clip's record came from the rank the partition names
            owner = {i: r for r, p in enumerate(parts) for i in p}
            assert all(int(row[4]) == owner[int(row[0])] for row in out.tolist())
            # ranks without a clip contributed nothing
            empty = [r for r, p in enumerate(parts) if not p]
            assert not set(empty) & set(int(v) for v in out[:, 4].tolist())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,lengths', [
    (4, [300, 20, 75]),                                   # 3 clips on 4 ranks: one rank holds none
    (4, [1824, 910, 777, 1500, 64, 333, 2048, 12, 905]),
    (8, [640, 1200, 87, 64, 333]),                        # 5 clips on 8 ranks: three ranks hold none
    (8, [1824, 910, 777, 1500, 64, 333, 2048, 12, 905, 640, 1200, 87, 1000, 5, 5, 5, 77, 431, 222, 90, 1403]),
    (8, []),                                              # no clip at all: every rank gathers an empty record set
])
def test_partition_and_gather_world4_and_world8_gloo_with_empty_ranks(world, lengths):
    """The clip-sharded evaluation's collectives at the node's real rank counts (3DPW-test has only a few dozen
    clips; H36M / a single video can have fewer clips than GPUs, so ranks with ZERO clips must take part in every collective)."""
    mp.spawn(_worker_many, args=(world, _free_port(), lengths), nprocs=world, join=True)
