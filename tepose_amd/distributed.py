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


# ms of ONE lock-step of tepose_amd.driver.run_clips (= one tepose_window_step call: pair projection, 2 recurrent launches per layer, layer-1
# projections, collapsed tail, SMPL) against the number of clips still active, published architecture (L = 2, H = 1024), seqlen 6, 1 x MI355X --
# measured points (profiles/r06_step_ms.json; bench.py `eval_driver.step_ms_table` re-measures them on the box it runs on); linear in between and beyond.
DEFAULT_STEP_MS = {1: 0.093, 2: 0.100, 5: 0.118, 10: 0.135, 19: 0.159, 37: 0.217, 64: 0.30}


class StepCost(object):
    """step_ms(B): piecewise-linear through measured (active clips, ms per lock-step) points; a MODEL, not a measurement."""

    def __init__(self, table=None):
        pts = sorted((int(b), float(ms)) for b, ms in (table or DEFAULT_STEP_MS).items())
        if not pts or pts[0][0] < 1:
            raise ValueError('step-cost table needs points at B >= 1')
        self.pts = pts

    def __call__(self, B):
        B = int(B)
        if B <= 0:
            return 0.0
        p = self.pts
        if len(p) == 1:
            return p[0][1]
        if B <= p[0][0]:
            lo, hi = p[0], p[1]
        elif B >= p[-1][0]:
            lo, hi = p[-2], p[-1]
        else:
            k = max(i for i in range(len(p) - 1) if p[i][0] <= B)
            lo, hi = p[k], p[k + 1]
        ms = lo[1] + (hi[1] - lo[1]) * (B - lo[0]) / float(hi[0] - lo[0])
        return max(ms, 0.5 * p[0][1])


def lockstep_seconds(lengths, seqlen, step_ms):
    """Predicted seconds of ONE rank advancing clips of these frame counts in lock-step (driver.run_clips; the reference's loop evaluate.py:247-269 and
    trainer.py:313-344 per clip): lock-step j serves the clips that still have a window j, sum_j step_ms(active(j)).  Windows of a clip are serial, so this is
    never below (longest clip's windows) x step_ms(1)."""
    w = sorted((max(int(n) - int(seqlen) + 1, 0) for n in lengths), reverse=True)
    total, prev = 0.0, 0
    # w[k] windows are served by k + 1 or more clips: steps w[k+1] .. w[k] - 1 have exactly k + 1 active clips
    for k in range(len(w) - 1, -1, -1):
        if w[k] > prev:
            total += (w[k] - prev) * step_ms(k + 1)
            prev = w[k]
    return total / 1e3


def partition_clips(lengths, world_size, step_ms=None, seqlen=6):
    """Clips -> ranks; returns a list of index lists, one per rank.  Deterministic, so every rank computes the same partition without communicating.
    step_ms=None (default; the synthetic weak-scaling bench): longest-processing-time greedy on FRAME totals -- clips sorted by frame count (ties by
    index), each goes to the currently lightest rank.
    step_ms=StepCost / callable (the clip-sharded evaluation): the executor advances a rank's clips in LOCK-STEP, so a rank's time is
    lockstep_seconds(its clips), dominated by its longest clip, not by its frame total.  Greedy by descending length on the predicted finish time: each
    clip goes to the rank whose predicted time AFTER taking it is smallest (ties: fewer frames, lower rank); the frame-LPT partition is kept when the
    model predicts no gain."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(lengths[i])
    if step_ms is None or world_size <= 1:
        return parts
    cost = lambda p: lockstep_seconds([lengths[i] for i in p], seqlen, step_ms)
    # Clips arrive in descending length, so a new clip is the shortest of its rank: for each of its w windows every clip already there is still active,
    # and taking it costs the rank  w x (step_ms(k + 1) - step_ms(k))  (k = clips of the rank that have windows at all).  O(clips x ranks), exact.
    mine = [[] for _ in range(world_size)]
    secs, frames, active = [0.0] * world_size, [0] * world_size, [0] * world_size
    for i in order:
        w = max(int(lengths[i]) - int(seqlen) + 1, 0)
        cand = [(secs[r] + w * (step_ms(active[r] + 1) - (step_ms(active[r]) if active[r] else 0.0)) / 1e3, frames[r], r) for r in range(world_size)]
        t, _, r = min(cand)
        mine[r].append(i)
        secs[r], frames[r], active[r] = t, frames[r] + int(lengths[i]), active[r] + (1 if w > 0 else 0)
    return mine if max(secs) < max(cost(p) for p in parts) else parts


def predicted_scaling(lengths, seqlen, step_ms, worlds=(1, 2, 4, 8)):
    """The cost model's view of clip-sharded evaluation (MODEL, not measurement): per world size the predicted makespan of the lock-step-aware
    partition and of frame-LPT, plus the critical path no sharding can beat -- the longest clip alone on a GPU."""
    w_max = max([max(int(n) - int(seqlen) + 1, 0) for n in lengths] + [0])
    out = {'model': 'sum over lock-steps of step_ms(active clips); NOT a measurement', 'critical_path_seconds': w_max * step_ms(1) / 1e3,
           'step_ms_points': [[b, ms] for b, ms in getattr(step_ms, 'pts', [])], 'predicted_seconds': {}, 'predicted_seconds_frame_lpt': {}}
    for n in worlds:
        for key, sm in (('predicted_seconds', step_ms), ('predicted_seconds_frame_lpt', None)):
            parts = partition_clips(lengths, n, sm, seqlen)
            out[key][str(n)] = max(lockstep_seconds([lengths[i] for i in p], seqlen, step_ms) for p in parts)
    one = out['predicted_seconds']['1']
    out['predicted_speedup'] = {k: (one / v if v > 0 else 1.0) for k, v in out['predicted_seconds'].items()}
    return out


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


def broadcast_model_weights(model, src=0):
    """The weight collective of SURVEY 8e: rank `src` holds the loaded model (TePose or the VIBE bootstrap), every other rank a
    freshly constructed one of the same architecture.  `src` packs (if it has not yet), then only the fp32 sections of the packed blob travel -- one broadcast
    per byte range of `Engine.fp32_ranges()` (header, packed matrices, biases, SMPL tables, collapsed maps: 276 of 770 MB at
    L = 2 / H = 1024) -- and every receiver rebuilds the hi / lo planes on its own GPU (`tepose_derive_planes`: byte-identical
    to the sender's blob).  Backend nccl = RCCL over xGMI.  Returns {'ms', 'bytes', 'blob_bytes'} (ms includes the receivers'
    plane derivation); with no initialised process group it only packs and reports ms = None."""
    import time
    eng = model._engine
    dev = next(model.parameters()).device
    if not (dist.is_available() and dist.is_initialized()):
        eng.pack_model(model, dev)
        return {'ms': None, 'bytes': 0, 'blob_bytes': eng.packed_bytes}
    rank = dist.get_rank()
    with torch.cuda.device(dev):
        if rank == src:
            eng.pack_model(model, dev)
            blob = eng.blob
        else:
            blob = torch.zeros(eng.packed_bytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        ranges = eng.fp32_ranges()
        for off, n in ranges:
            dist.broadcast(blob[off:off + n], src=src)
        if rank != src:
            eng.adopt_blob(blob, model, derive=True)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3
    return {'ms': ms, 'bytes': sum(n for _, n in ranges), 'blob_bytes': eng.packed_bytes}


def device_identity(device):
    """A string that is equal on two ranks exactly when they drive the same physical GPU: the device UUID where the runtime
    exposes one, else the PCI bus id."""
    props = torch.cuda.get_device_properties(device)
    uuid = getattr(props, 'uuid', None)
    if uuid is not None and str(uuid).strip('0-') != '':
        return 'uuid:%s' % uuid
    bus = getattr(props, 'pci_bus_id', None)
    dom = getattr(props, 'pci_domain_id', 0)
    dv = getattr(props, 'pci_device_id', 0)
    if bus is not None:
        return 'pci:%04x:%02x:%02x' % (int(dom), int(bus), int(dv))
    return 'index:%d' % torch.device(device).index


def count_distinct_devices(device):
    """all_gather of `device_identity` over the ranks -> (number of distinct GPUs, list of identities by rank).  An N-rank run on
    N GPUs reports N; N ranks sharing one GPU (the CPU-box test set-up) report 1."""
    me = device_identity(device)
    if not (dist.is_available() and dist.is_initialized()):
        return 1, [me]
    ids = [None] * dist.get_world_size()
    dist.all_gather_object(ids, me)
    return len(set(ids)), ids
