"""Clip-level data parallelism for the TePose hot path.

Independent units are *clips* (one `vid_name` group, reference evaluate.py:181-206,214);
windows inside a clip are serial (evaluate.py:247-269), so a clip is never split.  One
process per GPU; the data path has no collective.  Two collectives exist around it:
one broadcast of the packed-weight blob (rank 0 -> all) and one gather of fixed-size
per-clip records (all -> rank 0).  Backend `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the
CPU tests.
"""
import torch
import torch.distributed as dist


def partition_clips(lengths, world_size):
    """Longest-processing-time greedy: clips sorted by frame count (ties by index), each
    goes to the currently lightest rank.  Returns a list of index lists, one per rank.
    Deterministic, so every rank computes the same partition without communicating."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(lengths[i])
    return parts


def imbalance(lengths, parts):
    """max rank frames / mean rank frames (1.0 = perfect)."""
    loads = [sum(int(lengths[i]) for i in p) for p in parts]
    mean = sum(loads) / max(1, len(loads))
    return max(loads) / mean if mean > 0 else 1.0


def broadcast_blob(blob, src=0):
    """One broadcast of the packed weights (uint8 tensor, same size on every rank)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def gather_records(records, dst=0):
    """records: [n_local, k] float64 tensor of per-clip rows (e.g. clip_id, n_frames,
    sum_mpjpe, ...).  Ranks may hold different n_local: rows are padded to the maximum and
    the true counts travel in a first all_gather.  Returns the concatenated [sum n, k]
    tensor on `dst`, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()):
        return records               # (an initialised group of ONE rank runs the collectives: `--force-dist`)
    world, rank = dist.get_world_size(), dist.get_rank()
    k = records.shape[1]
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = int(max(int(c.item()) for c in counts))
    padded = torch.zeros(nmax, k, dtype=records.dtype, device=records.device)
    padded[:records.shape[0]] = records
    bufs = [torch.zeros_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:int(c.item())] for b, c in zip(bufs, counts)], dim=0)
