"""End-to-end clip-sharded evaluation (the reference's evaluate.py:209-462 flow) on MI355X.

    python tools/evaluate_clips.py                       # synthetic 3DPW-like database, 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 tools/evaluate_clips.py
    python tools/evaluate_clips.py --cfg configs/repr_wpw_3dpw_model.yaml --dataset 3dpw \
        --base-data data/base_data --db-dir data/preprocessed_data --vibe-ckpt data/vibe_data/vibe_model_wo_3dpw.pth.tar
    python tools/evaluate_clips.py --db <x_db.pt> --pseudotheta <x_pseudotheta.pt> --base-data data/base_data \
        --ckpt <tepose.pth.tar> --vibe-ckpt <vibe.pth.tar> [--layers 2 --hidden 1024]
    python tools/evaluate_clips.py --check --cfg configs/repr_wpw_3dpw_model.yaml --dataset 3dpw --base-data data/base_data \
        --db-dir data/preprocessed_data --vibe-ckpt data/vibe_data/vibe_model_wo_3dpw.pth.tar
        # real-data readiness, no GPU: which licence-gated files (reference README.md:21-25,37-62) are there, their shapes, what is
        # missing, and the published row a run on them will be compared with

Real data (`--db` / `--db-dir`): every table comes from the user's files as in the reference -- `J_regressor_h36m.npy`,
`smpl_mean_params.npz`, `SMPL_NEUTRAL.pkl`, `J_regressor_extra.npy` from `--base-data` (evaluate.py:109,130-135), the
architecture from `--cfg` (MODEL.TGRU.*, lib/core/config.py:123-126), the window from evaluate.py:141 (seqlen 6), the
database names from evaluate.py:146-162, the TePose checkpoint from `--ckpt` or the config's TRAIN.PRETRAINED
(evaluate.py:121-127 exits when it is no file: so does this tool), the bootstrap VIBE (2 layers, 1024 hidden,
evaluate.py:93-101) from `--vibe-ckpt`.  Nothing synthetic is mixed in; a missing input is refused, not replaced.

Without a database the run is the throughput harness: random-init weights of the published architecture and a
synthetic database with the `*_db.pt` schema.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import synth  # noqa: E402
from tepose_amd.config import EVAL_SEQLEN, compare_with_published, eval_db_paths, published_row, update_cfg  # noqa: E402
from tepose_amd.data import (load_base_data, load_eval_db, load_generator_state_dict, split_db_into_clips,  # noqa: E402
                             synthetic_eval_db)
from tepose_amd.distributed import StepCost, imbalance, partition_clips, predicted_scaling  # noqa: E402
from tepose_amd.evaluate import evaluate_clips, gather_and_reduce, gather_rank_stats  # noqa: E402
from tepose_amd.smpl import SMPL  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402
from tepose_amd.vibe import VIBE  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', help="the reference's experiment YAML (configs/repr_*.yaml): architecture, TITLE, TRAIN.PRETRAINED")
    ap.add_argument('--db'); ap.add_argument('--pseudotheta'); ap.add_argument('--ckpt'); ap.add_argument('--vibe-ckpt')
    ap.add_argument('--db-dir', help='directory of the preprocessed databases: file names derived as evaluate.py:146-162 does')
    ap.add_argument('--base-data', help='directory with J_regressor_h36m.npy, smpl_mean_params.npz, SMPL_NEUTRAL.pkl, '
                    'J_regressor_extra.npy (required with --db / --db-dir)')
    ap.add_argument('--dataset', default='3dpw', choices=['3dpw', 'h36m', 'mpii3d'])
    ap.add_argument('--seq', default='', help='evaluate only clips whose name contains this (evaluate.py --seq)')
    ap.add_argument('--seqlen', type=int, default=None, help='window length (default: 6, evaluate.py:141)')
    ap.add_argument('--clips', type=int, default=37, help='synthetic database: number of clips')
    ap.add_argument('--min-len', type=int, default=300); ap.add_argument('--max-len', type=int, default=1800)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--share-device0', action='store_true', help='testing only: every rank uses cuda:0 (with --backend gloo)')
    ap.add_argument('--gpus', type=int, default=0, help='start this many ranks (one per GPU) from a plain `python` call')
    ap.add_argument('--layers', type=int, default=None); ap.add_argument('--hidden', type=int, default=None)
    ap.add_argument('--vibe-layers', type=int, default=None, help='bootstrap model (default: 2 / 1024 as evaluate.py:93-101 '
                    'with real data, the TePose architecture in the synthetic harness)')
    ap.add_argument('--vibe-hidden', type=int, default=None)
    ap.add_argument('--as-rank', type=int, default=-1, help='debug: process the share of this rank of --of-world in ONE process')
    ap.add_argument('--of-world', type=int, default=2)
    ap.add_argument('--filter', action='store_true', help='evaluate.py --filter: slerp-smooth the predicted rotations per clip, re-pose SMPL, evaluate the '
                    'joints of that mesh (evaluate.py:273-291); not for mpii3d (the reference\'s branch needs the J_regressor)')
    ap.add_argument('--check', action='store_true', help='real-data readiness report (CPU only, one JSON line): found / missing files with their shapes, '
                    'the published row a run will be compared with; exit code 0 = everything a real-data run needs is there')
    ap.add_argument('--force-dist', action='store_true', help='world size 1: still initialise the process group (nccl = RCCL) and '
                    'run the barrier / all_reduce / gathers, so that the N-GPU collectives execute on a 1-GPU box')
    return ap.parse_args(argv)


def resolve_plan(args):
    """Everything the run derives from its arguments before a GPU is touched (CPU-testable): architecture, window,
    database / checkpoint paths, and the refusals of an incomplete real-data request.  Returns a dict."""
    cfg = update_cfg(args.cfg) if args.cfg else None
    plan = {'real': bool(args.db or args.db_dir), 'dataset': args.dataset, 'cfg_title': cfg.TITLE if cfg else None}
    plan['layers'] = args.layers if args.layers is not None else (int(cfg.MODEL.TGRU.NUM_LAYERS) if cfg else 2)
    plan['hidden'] = args.hidden if args.hidden is not None else (int(cfg.MODEL.TGRU.HIDDEN_SIZE) if cfg else 1024)
    plan['seqlen'] = args.seqlen if args.seqlen is not None else EVAL_SEQLEN            # evaluate.py:141 overrides DATASET.SEQLEN
    plan['ckpt'] = args.ckpt or (cfg.TRAIN.PRETRAINED if cfg and cfg.TRAIN.PRETRAINED else None)
    plan['vibe_ckpt'] = args.vibe_ckpt
    if plan['real']:
        plan['vibe_layers'] = args.vibe_layers if args.vibe_layers is not None else 2     # evaluate.py:93-101
        plan['vibe_hidden'] = args.vibe_hidden if args.vibe_hidden is not None else 1024
        if args.db:
            if not args.pseudotheta:
                raise SystemExit('--db needs --pseudotheta (evaluate.py:149,174 reads both files)')
            plan['db'], plan['pseudotheta'] = args.db, args.pseudotheta
        else:
            if cfg is None and args.dataset == 'h36m':
                raise SystemExit('--db-dir with --dataset h36m needs --cfg: the file name depends on the config TITLE (evaluate.py:149-155)')
            plan['db'], plan['pseudotheta'] = eval_db_paths(cfg if cfg else update_cfg_default(), args.dataset, args.db_dir)
        if not args.base_data:
            raise SystemExit('real data needs --base-data DIR (J_regressor_h36m.npy, smpl_mean_params.npz, SMPL_NEUTRAL.pkl, '
                             'J_regressor_extra.npy): synthetic stand-ins would make every metric meaningless')
        plan['base_data'] = args.base_data
        for what, path in (('database', plan['db']), ('pseudo-theta file', plan['pseudotheta'])):
            if not os.path.isfile(str(path)):
                raise SystemExit('%s %r does not exist' % (what, str(path)))
        if not plan['ckpt'] or not os.path.isfile(str(plan['ckpt'])):
            raise SystemExit('%r is not a pretrained model! (evaluate.py:121-127; give --ckpt or a --cfg whose TRAIN.PRETRAINED exists)'
                             % (plan['ckpt'],))
        if not plan['vibe_ckpt'] or not os.path.isfile(str(plan['vibe_ckpt'])):
            raise SystemExit('real data needs --vibe-ckpt (the bootstrap model of evaluate.py:89-107 predicts frames 0..T-2)')
    else:
        plan['vibe_layers'] = args.vibe_layers if args.vibe_layers is not None else plan['layers']
        plan['vibe_hidden'] = args.vibe_hidden if args.vibe_hidden is not None else plan['hidden']
    return plan


def _describe_file(path, kind):
    """{'path', 'exists', ...shapes} of one input file, read on the CPU; a file that exists but does not parse says so in 'error'."""
    d = {'path': str(path) if path else None, 'kind': kind, 'exists': bool(path) and os.path.isfile(str(path))}
    if not d['exists']:
        return d
    d['bytes'] = os.path.getsize(str(path))
    try:
        if kind == 'npy':
            d['shape'] = list(np.load(str(path), mmap_mode='r').shape)
        elif kind == 'npz':
            z = np.load(str(path))
            d['arrays'] = {k: list(z[k].shape) for k in z.files}
        elif kind == 'smpl_pkl':
            from tepose_amd.smpl import load_smpl_pkl
            d['arrays'] = {k: list(np.asarray(v).shape) for k, v in load_smpl_pkl(str(path)).items()}
        elif kind == 'db':
            import joblib
            db = joblib.load(str(path))
            if isinstance(db, dict):
                d['arrays'] = {k: list(np.asarray(v).shape) for k, v in db.items()}
                if 'vid_name' in db:
                    d['clips'] = int(len(np.unique(np.asarray(db['vid_name']))))
                    d['frames'] = int(len(db['vid_name']))
            else:
                d['shape'] = list(np.asarray(db).shape)
        elif kind == 'checkpoint':
            from tepose_amd.data import load_checkpoint
            ck = load_checkpoint(str(path))
            sd = ck['gen_state_dict'] if 'gen_state_dict' in ck else ck
            d['tensors'] = len(sd)
            d['parameters'] = int(sum(int(np.prod(tuple(v.shape))) for v in sd.values() if hasattr(v, 'shape')))
            if 'performance' in ck:
                d['performance'] = float(ck['performance'])
    except Exception as e:                                         # noqa: BLE001 -- the report names the problem instead of dying on it
        d['error'] = '%s: %s' % (type(e).__name__, e)
    return d


def readiness_report(args):
    """What a real-data run of this tool needs (the licence-gated downloads of the reference's README.md:21-25,37-62) against what is on disk.  CPU only."""
    cfg = update_cfg(args.cfg) if args.cfg and os.path.isfile(args.cfg) else None
    files = []
    if args.cfg:
        files.append({'path': args.cfg, 'kind': 'cfg', 'exists': os.path.isfile(args.cfg), 'title': cfg.TITLE if cfg else None})
    base = args.base_data or 'data/base_data'
    files += [_describe_file(os.path.join(base, 'J_regressor_h36m.npy'), 'npy'), _describe_file(os.path.join(base, 'J_regressor_extra.npy'), 'npy'),
              _describe_file(os.path.join(base, 'smpl_mean_params.npz'), 'npz'), _describe_file(os.path.join(base, 'SMPL_NEUTRAL.pkl'), 'smpl_pkl')]
    db = pse = None
    if args.db:
        db, pse = args.db, args.pseudotheta
    elif args.dataset == 'h36m' and cfg is None:
        files.append({'path': None, 'kind': 'db', 'exists': False, 'error': '--dataset h36m needs --cfg: the database name depends on the config TITLE (evaluate.py:149-155)'})
    else:
        db, pse = eval_db_paths(cfg if cfg else update_cfg_default(), args.dataset, args.db_dir or 'data/preprocessed_data')
    if db:
        files += [_describe_file(db, 'db'), _describe_file(pse, 'db')]
    ckpt = args.ckpt or (cfg.TRAIN.PRETRAINED if cfg and cfg.TRAIN.PRETRAINED else None)
    files += [_describe_file(ckpt, 'checkpoint'), _describe_file(args.vibe_ckpt or 'data/vibe_data/vibe_model_wo_3dpw.pth.tar', 'checkpoint')]
    want = {'J_regressor_h36m.npy': [17, 6890], 'J_regressor_extra.npy': [9, 6890]}
    for f in files:
        w = want.get(os.path.basename(str(f.get('path'))))
        if w and f.get('exists') and f.get('shape') != w:
            f['error'] = 'shape %s, want %s' % (f.get('shape'), w)
    missing = [f['path'] for f in files if not f.get('exists')]
    broken = [f['path'] for f in files if f.get('exists') and f.get('error')]
    title = cfg.TITLE if cfg else None
    return {'ready': not missing and not broken, 'dataset': args.dataset, 'cfg_title': title, 'files': files, 'missing': missing, 'unreadable': broken,
            'published': published_row(title, args.dataset),
            'note': 'published = the reference\'s own accuracy row for this config / evaluation set (BASELINE.md section 1; asset/wpw.png, asset/wopw.png); a real-data '
                    'run prints it next to the measured metrics with the differences'}


def update_cfg_default():
    from tepose_amd.config import get_cfg_defaults
    return get_cfg_defaults()


def load_assets(plan):
    """SMPL tables, mean parameters, the H36M joint regressor and the clips: from the user's files (real data) or from
    tepose_amd.synth (throughput harness) -- never a mix.  CPU only."""
    if plan['real']:
        base = load_base_data(plan['base_data'])
        clips = load_eval_db(plan['db'], plan['pseudotheta'], target_action=plan.get('seq', ''))
        return {'smpl_tables': base['smpl_tables'], 'mean_tepose': base['mean_params'], 'mean_vibe': base['mean_params'],
                'J_regressor_h36m': base['J_regressor_h36m'], 'clips': clips, 'source': 'files'}
    smpl_np = synth.synthetic_smpl(0)
    vstate = synth.synthetic_vibe_state_dict(plan['vibe_layers'], plan['vibe_hidden'], 1)
    mean_v = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0],
              'cam': vstate['regressor.init_cam'][0]}
    return {'smpl_tables': smpl_np, 'mean_tepose': synth.synthetic_mean_params(0), 'mean_vibe': mean_v,
            'J_regressor_h36m': smpl_np['J_regressor_h36m'], 'vibe_state': vstate, 'clips': None, 'source': 'synth'}


def main():
    args = parse_args()
    if args.check:
        rep = readiness_report(args)
        print(json.dumps(rep))
        raise SystemExit(0 if rep['ready'] else 1)
    plan = resolve_plan(args)
    plan['seq'] = args.seq
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
    T = plan['seqlen']
    assets = load_assets(plan)
    smpl_np = assets['smpl_tables']
    if plan['real']:
        clips = assets['clips']
    else:
        lens = (args.min_len + (args.max_len - args.min_len) * synth.uniform01('evalclips', args.clips)).astype(int)
        db, pse = synthetic_eval_db(list(lens), seed=0)
        clips = split_db_into_clips(db, pse)
    # Weights: rank 0 loads (checkpoint files or the synthetic state dicts) and packs; every other rank constructs the bare
    # architecture and adopts rank 0's blob from the RCCL broadcast (tepose_amd.distributed.broadcast_model_weights: the fp32
    # sections travel, the hi / lo planes are re-derived per GPU) -- the collective SURVEY 8e / the north star name.
    from tepose_amd.tepose import TePose

    def bare_models():
        m = TePose(seqlen=T, n_layers=plan['layers'], hidden_size=plan['hidden'], pretrained='',
                   smpl=SMPL.from_tables(smpl_np), smpl_mean_params=assets['mean_tepose']).to(dev).eval()
        v = VIBE(seqlen=T, n_layers=plan['vibe_layers'], hidden_size=plan['vibe_hidden'], add_linear=True, use_residual=True,
                 pretrained='', smpl=SMPL.from_tables(smpl_np), smpl_mean_params=assets['mean_vibe'])
        return m, v
    if plan['real']:
        model, vibe = bare_models()
        if rank == 0 or not use_dist:
            model.load_state_dict(load_generator_state_dict(plan['ckpt']), strict=True)           # evaluate.py:121-124
            vibe.load_state_dict(load_generator_state_dict(plan['vibe_ckpt']), strict=False)      # evaluate.py:103-105
    elif rank == 0 or not use_dist:
        model, _, _ = build_model(plan['layers'], plan['hidden'], seed=0, device=dev, smpl_np=smpl_np, seqlen=T)
        _, vibe = bare_models()
        sd = vibe.state_dict()
        for k, v in assets['vibe_state'].items():
            sd[k] = torch.from_numpy(v)
        vibe.load_state_dict(sd)
        if args.ckpt:
            model.load_state_dict(load_generator_state_dict(args.ckpt), strict=True)
        if args.vibe_ckpt:
            vibe.load_state_dict(load_generator_state_dict(args.vibe_ckpt), strict=False)
    else:
        model, vibe = bare_models()
    vibe = vibe.to(dev).eval()
    bcast = None
    if use_dist:
        from tepose_amd.distributed import broadcast_model_weights, count_distinct_devices
        b1 = broadcast_model_weights(model, src=0)
        b2 = broadcast_model_weights(vibe, src=0)
        n_seen, rank_devs = count_distinct_devices(dev)
        bcast = {'weight_broadcast_ms': b1['ms'] + b2['ms'], 'weight_broadcast_MB': (b1['bytes'] + b2['bytes']) / 1e6,
                 'weight_blob_MB': (b1['blob_bytes'] + b2['blob_bytes']) / 1e6, 'n_ranks_seen': n_seen, 'rank_devices': rank_devs}
    J = torch.from_numpy(assets['J_regressor_h36m']) if args.dataset != 'mpii3d' else None      # evaluate.py:109,203
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    if args.as_rank >= 0:
        recs, mine = evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset=args.dataset, rank=args.as_rank, world=args.of_world, avg_filter=args.filter)
    else:
        recs, mine = evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset=args.dataset, rank=rank, world=world, avg_filter=args.filter)
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
               'imbalance_max_over_mean_rank_frames': imbalance(lens, partition_clips(lens, world, StepCost(), T)),
               # the lock-step cost model's view (MODEL, not measurement): a rank's time is its longest clip's chain of window steps, so clip-sharding a
               # database whose longest clip dominates cannot scale like the clip count -- predicted makespans for 1 / 2 / 4 / 8 GPUs and the floor
               'lockstep_cost_model': predicted_scaling(lens, T, StepCost()),
               'per_rank': stats,      # seconds, clips, frames and longest clip (= serial window chain) of every rank
               'frames_per_s': frames / float(el.item()), 'metrics_mm': res, 'avg_filter': bool(args.filter),
               'data': 'real' if plan['real'] else 'synthetic db + random-init weights (metric values are meaningless)',
               'tables': assets['source'], 'arch': {'layers': plan['layers'], 'hidden': plan['hidden'], 'cfg': plan['cfg_title']}}
        if plan['real']:
            # the reference's own accuracy row for this config / evaluation set next to what was measured
            out['vs_published'] = compare_with_published(res, plan['cfg_title'], args.dataset)
        if use_dist:
            out['dist_backend'] = dist.get_backend()
            out.update(bcast)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
