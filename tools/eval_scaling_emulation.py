"""Clip-sharded evaluation at world sizes 1 / 2 / 4 / 8, EMULATED on one GPU: every rank's share of the partition is evaluated alone, one rank after the
other, and timed -- on a real node the ranks run on their own GPUs with no data-path communication, so the makespan of the N-GPU run is
the slowest rank's time (plus the one-off weight broadcast and record gather, not included here).  Prints the measured makespans next to the lock-step cost
model's predictions (tepose_amd.distributed.predicted_scaling, fed with the step-cost table measured on this box).

    python tools/eval_scaling_emulation.py [--clips 37] [--min-len 300] [--max-len 1800]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import synth  # noqa: E402
from tepose_amd.data import split_db_into_clips, synthetic_eval_db  # noqa: E402
from tepose_amd.distributed import StepCost, partition_clips, predicted_scaling  # noqa: E402
from tepose_amd.evaluate import evaluate_clips, measure_step_ms  # noqa: E402
from tepose_amd.smpl import SMPL  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402
from tepose_amd.vibe import VIBE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=37)
    ap.add_argument('--min-len', type=int, default=300)
    ap.add_argument('--max-len', type=int, default=1800)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    T, L, H = 6, 2, 1024
    smpl_np = synth.synthetic_smpl(0)
    lens = (args.min_len + (args.max_len - args.min_len) * synth.uniform01('evalclips', args.clips)).astype(int)
    db, pse = synthetic_eval_db(list(lens), seed=0)
    clips = split_db_into_clips(db, pse)
    model, _, _ = build_model(L, H, seed=0, device=dev, smpl_np=smpl_np, seqlen=T)
    vstate = synth.synthetic_vibe_state_dict(L, H, 1)
    mean_v = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0], 'cam': vstate['regressor.init_cam'][0]}
    vibe = VIBE(seqlen=T, n_layers=L, hidden_size=H, add_linear=True, use_residual=True, pretrained='', smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean_v)
    sd = vibe.state_dict()
    for k, v in vstate.items():
        sd[k] = torch.from_numpy(v)
    vibe.load_state_dict(sd)
    vibe = vibe.to(dev).eval()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    table = measure_step_ms(model, T, J_regressor=J)
    cost = StepCost(table)
    lengths = [len(c['features']) for c in clips.values()]
    out = {'clips': len(lengths), 'frames': int(sum(lengths)), 'longest_clip_frames': int(max(lengths)), 'seqlen': T,
           'how': 'every rank of every world size evaluated alone on ONE GPU, one after the other; makespan = the slowest rank (no broadcast / gather time)',
           'step_ms_table': {str(k): v for k, v in table.items()}, 'model': predicted_scaling(lengths, T, cost), 'measured': {}}
    evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset='3dpw', step_ms=cost)           # warm-up: packs, sizes the workspaces
    for world in (1, 2, 4, 8):
        parts = partition_clips(lengths, world, cost, T)
        per_rank = []
        for r in range(world):
            best = None
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset='3dpw', rank=r, world=world, step_ms=cost)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            per_rank.append(best)
        out['measured'][str(world)] = {'makespan_seconds': max(per_rank), 'per_rank_seconds': per_rank, 'clips_per_rank': [len(p) for p in parts],
                                       'frames_per_rank': [int(sum(lengths[i] for i in p)) for p in parts]}
    one = out['measured']['1']['makespan_seconds']
    out['measured_speedup'] = {w: one / v['makespan_seconds'] for w, v in out['measured'].items()}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
