"""End-to-end clip-sharded evaluation (the reference's evaluate.py:209-462 flow) on MI355X.

    python tools/evaluate_clips.py                       # synthetic 3DPW-like database, 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 tools/evaluate_clips.py
    python tools/evaluate_clips.py --db data/preprocessed_data/3dpw_test_db.pt \
        --pseudotheta data/preprocessed_data/3dpw_test_pseudotheta.pt --ckpt <tepose.pth.tar> --vibe-ckpt <vibe.pth.tar>

Real files are licence-gated and absent from this repo; without them the run uses random-init
weights of the published architecture (n_layers=2, hidden=1024, seqlen=6 as evaluate.py:141
hard-codes) and a synthetic database with the `*_db.pt` schema, so the numbers that mean
something are the throughput ones.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import synth  # noqa: E402
from tepose_amd.data import load_eval_db, load_generator_state_dict, split_db_into_clips, synthetic_eval_db  # noqa: E402
from tepose_amd.distributed import imbalance, partition_clips  # noqa: E402
from tepose_amd.evaluate import evaluate_clips, gather_and_reduce, gather_rank_stats  # noqa: E402
from tepose_amd.smpl import SMPL  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402
from tepose_amd.vibe import VIBE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--db'); ap.add_argument('--pseudotheta'); ap.add_argument('--ckpt'); ap.add_argument('--vibe-ckpt')
    ap.add_argument('--dataset', default='3dpw', choices=['3dpw', 'h36m', 'mpii3d'])
    ap.add_argument('--seqlen', type=int, default=6)
    ap.add_argument('--clips', type=int, default=37, help='synthetic database: number of clips')
    ap.add_argument('--min-len', type=int, default=300); ap.add_argument('--max-len', type=int, default=1800)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--share-device0', action='store_true', help='testing only: every rank uses cuda:0 (with --backend gloo)')
    ap.add_argument('--gpus', type=int, default=0, help='start this many ranks (one per GPU) from a plain `python` call')
    ap.add_argument('--layers', type=int, default=2); ap.add_argument('--hidden', type=int, default=1024)
    ap.add_argument('--as-rank', type=int, default=-1, help='debug: process the share of this rank of --of-world in ONE process')
    ap.add_argument('--of-world', type=int, default=2)
    ap.add_argument('--force-dist', action='store_true', help='world size 1: still initialise the process group (nccl = RCCL) and '
                    'run the barrier / all_reduce / gathers, so that the N-GPU collectives execute on a 1-GPU box')
    args = ap.parse_args()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # before the first HIP call (dmabuf IPC only on this pool)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # GPU-free parent: the ranks are children of the stock launcher (never an exec of a process that touched HIP)
        import socket
        import subprocess
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ); env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0'); env.setdefault('OMP_NUM_THREADS', '8')
        raise SystemExit(subprocess.call(
            [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
             '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = 0 if args.share_device0 else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            import socket
            sk = socket.socket(); sk.bind(('127.0.0.1', 0))
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(sk.getsockname()[1]), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
            sk.close()
        dist.init_process_group(args.backend, device_id=dev) if args.backend == 'nccl' else dist.init_process_group(args.backend)
    T = args.seqlen
    smpl_np = synth.synthetic_smpl(0)
    if args.db:
        clips = load_eval_db(args.db, args.pseudotheta)
    else:
        lens = (args.min_len + (args.max_len - args.min_len) * synth.uniform01('evalclips', args.clips)).astype(int)
        db, pse = synthetic_eval_db(list(lens), seed=0)
        clips = split_db_into_clips(db, pse)
    # Weights: rank 0 loads (checkpoint files or the synthetic state dicts) and packs; every other rank constructs the bare
    # architecture and adopts rank 0's blob from the RCCL broadcast (tepose_amd.distributed.broadcast_model_weights: the fp32
    # sections travel, the hi / lo planes are re-derived per GPU) -- the collective SURVEY 8e / the north star name.
    mean_t = synth.synthetic_mean_params(0)
    vstate = synth.synthetic_vibe_state_dict(args.layers, args.hidden, 1)
    mean_v = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0],
              'cam': vstate['regressor.init_cam'][0]}
    if rank == 0 or not use_dist:
        model, _, _ = build_model(args.layers, args.hidden, seed=0, device=dev, smpl_np=smpl_np, seqlen=T)
        vibe = VIBE(seqlen=T, n_layers=args.layers, hidden_size=args.hidden, add_linear=True, use_residual=True, pretrained='',
                    smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean_v)
        sd = vibe.state_dict()
        for k, v in vstate.items():
            sd[k] = torch.from_numpy(v)
        vibe.load_state_dict(sd)
        if args.ckpt:
            model.load_state_dict(load_generator_state_dict(args.ckpt), strict=True)          # evaluate.py:121-124
        if args.vibe_ckpt:
            vibe.load_state_dict(load_generator_state_dict(args.vibe_ckpt), strict=False)     # evaluate.py:103-105
    else:
        from tepose_amd.tepose import TePose
        model = TePose(seqlen=T, n_layers=args.layers, hidden_size=args.hidden, pretrained='', smpl=SMPL.from_tables(smpl_np),
                       smpl_mean_params=mean_t).to(dev).eval()
        vibe = VIBE(seqlen=T, n_layers=args.layers, hidden_size=args.hidden, add_linear=True, use_residual=True, pretrained='',
                    smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean_v)
    vibe = vibe.to(dev).eval()
    bcast = None
    if use_dist:
        from tepose_amd.distributed import broadcast_model_weights, count_distinct_devices
        b1 = broadcast_model_weights(model, src=0)
        b2 = broadcast_model_weights(vibe, src=0)
        n_seen, rank_devs = count_distinct_devices(dev)
        bcast = {'weight_broadcast_ms': b1['ms'] + b2['ms'], 'weight_broadcast_MB': (b1['bytes'] + b2['bytes']) / 1e6,
                 'weight_blob_MB': (b1['blob_bytes'] + b2['blob_bytes']) / 1e6, 'n_ranks_seen': n_seen, 'rank_devices': rank_devs}
    J = torch.from_numpy(smpl_np['J_regressor_h36m']) if args.dataset != 'mpii3d' else None
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    if args.as_rank >= 0:
        recs, mine = evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset=args.dataset, rank=args.as_rank, world=args.of_world)
    else:
        recs, mine = evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset=args.dataset, rank=rank, world=world)
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0
    el = torch.tensor([mine_s], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    res = gather_and_reduce(recs)
    stats = gather_rank_stats(rank, mine_s, [len(c['features']) for c in clips.values()], mine)
    if rank == 0:
        frames = int(sum(len(c['features']) for c in clips.values()))
        lens = [len(c['features']) for c in clips.values()]
        out = {'clips': len(clips), 'frames': frames, 'seqlen': T, 'n_gpus': world, 'seconds': float(el.item()),
               'imbalance_max_over_mean_rank_frames': imbalance(lens, partition_clips(lens, world)),
               'per_rank': stats,      # seconds, clips, frames and longest clip (= serial window chain) of every rank
               'frames_per_s': frames / float(el.item()), 'metrics_mm': res,
               'data': 'real' if args.db else 'synthetic db + random-init weights (metric values are meaningless)'}
        if use_dist:
            out['dist_backend'] = dist.get_backend()
            out.update(bcast)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
