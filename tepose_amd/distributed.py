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
