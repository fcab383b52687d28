"""CPU, world_size 2 on gloo: the clip partitioner, the blob broadcast and the padded
per-clip record gather used by the multi-GPU path (one process per GPU on the real node)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tepose_amd.distributed import broadcast_blob, gather_records, imbalance, partition_clips


def test_partition_is_a_balanced_exact_cover():
    lengths = [1824, 910, 777, 1500, 64, 333, 2048, 12, 905, 640, 1200, 87]
    for world in (1, 2, 4, 8):
        parts = partition_clips(lengths, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(lengths)))          # every clip exactly once, never split
        assert imbalance(lengths, parts) < 1.6
    assert partition_clips(lengths, 8) == partition_clips(list(lengths), 8)    # deterministic
    assert partition_clips([], 4) == [[], [], [], []]


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
