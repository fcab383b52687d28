"""Clip-sharded evaluation: the reference's evaluate.py:209-462 flow on one or more GPUs.

Per clip (evaluate.py:214-269): frames 0..T-2 come from the VIBE bootstrap model run on the
first T frames, frames T-1.. from the autoregressive TePose windows whose theta slots start
from the pseudo-theta file; then joints are converted / pelvis-aligned and MPJPE, PA-MPJPE,
acceleration error and MPVPE are taken per frame (evaluate.py:394-457).  Clips are independent:
each rank processes its share (tepose_amd.distributed.partition_clips), all its clips in
lock-step, and rank 0 gathers one fixed-size record per clip.
"""
import numpy as np
import torch

from . import distributed as D
from . import metrics as M
from .driver import run_clips


def _to_device(a, dev, n=None):
    """Host array (numpy or CPU tensor, any float type) -> contiguous fp32 device tensor of its first n rows.  The conversion runs in
    numpy on the calling thread: a torch CPU op on a few hundred KB wakes the whole intra-op thread pool, and on a host whose CPU quota
    is smaller than its core count (containers) the spinning pool gets the process throttled for the rest of the scheduler period --
    measured as random 40-90 ms stalls, several per evaluation (profiles/r05_eval_stalls.txt)."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)
    if n is not None:
        a = a[:n]
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def mpii3d_valid_map(valid_i, n_pred):
    """evaluate.py:397-405: indices of the frames whose `valid_i` flag is set, without those past the predictions
    (the reference trims them from the tail one at a time; the map is ascending, so that is a `< n_pred` filter).
    Host-side numpy / torch-CPU in, LongTensor out; empty = the clip is skipped ("No valid frames")."""
    vm = torch.as_tensor(valid_i).reshape(len(valid_i), -1)[:, 0].nonzero()[:, 0]
    return vm[vm < int(n_pred)]


@torch.no_grad()
def clip_metric_record(model, clip_id, clip, pred_j3d, pred_verts, dataset='3dpw'):
    """The per-clip metric block of evaluate.py:394-457 on device tensors: joint conversion (`convert_kps` spin ->
    mpii3d_test / common as index tables), the MPI-INF-3DHP `valid_i` frame filter, pelvis (joint -3 for mpii3d, mean of
    joints 2, 3 otherwise), MPJPE / PA-MPJPE / acceleration error per frame, MPVPE against SMPL(target theta) for 3DPW.
    pred_j3d [N, 49 | 14, 3], pred_verts [N, 6890, 3] = bootstrap frames + window predictions of ONE clip.
    Returns the clip's [8] record (metrics.clip_record) or None when the reference skips the clip."""
    dev = pred_j3d.device
    target = _to_device(clip['joints3D'], dev, pred_j3d.shape[0])                                      # evaluate.py:302
    valid_map = None
    if dataset == 'mpii3d':
        idx = torch.tensor(M.SPIN_TO_MPII3D_TEST, device=dev)
        target, pred_j3d = target[:, idx], pred_j3d[:, idx]
        valid_map = mpii3d_valid_map(clip['valid_i'], pred_j3d.shape[0]).to(dev)
        if valid_map.numel() == 0:
            return None
    elif target.shape[1] == 49:
        target = target[:, torch.tensor(M.SPIN_TO_COMMON, device=dev)]
    m = M.joint_metrics(pred_j3d, target, 'mpii3d' if dataset == 'mpii3d' else 'lsp')
    mpvpe = None
    if dataset == '3dpw':                                                   # evaluate.py:454-455
        n = pred_verts.shape[0]
        pose, shape = np.asarray(clip['pose'], dtype=np.float32)[:n], np.asarray(clip['shape'], dtype=np.float32)[:n]
        tt = _to_device(np.concatenate([np.zeros((len(pose), 3), np.float32), pose, shape], axis=1), dev)
        mpvpe = M.vertex_metric(pred_verts, M.gt_vertices(model, tt))
    return M.clip_record(clip_id, m, mpvpe=mpvpe, valid_map=valid_map)


@torch.no_grad()
def evaluate_clips(model, model_vibe, clips, seqlen, J_regressor=None, dataset='3dpw', rank=0, world=1, step_ms=None, avg_filter=False):
    """clips: OrderedDict name -> dict(features[N,2048], joints3D[N,J,3], theta_pseu[N,85], pose, shape).
    avg_filter: evaluate.py --filter (lines 273-291): per clip, the predicted rotations (bootstrap frames + window predictions) are slerp-smoothed
    (smooth_pose_mat, ratio 0.3), SMPL re-posed with the predicted betas, and the evaluated joints are the H36M regressor's 14 joints of THAT mesh; MPVPE
    keeps the unfiltered vertices, as the reference does (line 299).  Needs a J_regressor (the reference's branch indexes it: no mpii3d).
    Returns ([n_local_clips, 8] float64 record tensor, list of local clip indices)."""
    dev = next(model.parameters()).device
    T = int(seqlen)
    names = list(clips.keys())
    lengths = [len(clips[n]['features']) for n in names]
    # the executor below advances a rank's clips in lock-step: partition on the lock-step cost model, not on frame totals (distributed.partition_clips)
    mine = D.partition_clips(lengths, world, step_ms=step_ms if step_ms is not None else D.StepCost(), seqlen=T)[rank]
    mine = [i for i in mine if lengths[i] >= T]                       # evaluate.py:226-227
    if not mine:
        return torch.zeros(0, 8, dtype=torch.float64, device=dev), mine
    feats = [_to_device(clips[names[i]]['features'], dev) for i in mine]
    inits = [_to_device(clips[names[i]]['theta_pseu'], dev, T - 1) for i in mine]
    # bootstrap: VIBE over the first T frames of every clip, keep frames 0..T-2 (evaluate.py:233-245)
    if avg_filter and J_regressor is None:
        raise ValueError('avg_filter regresses the joints of the re-posed mesh with J_regressor (evaluate.py:289-291): give one')
    boot = model_vibe(torch.stack([f[:T] for f in feats]), J_regressor=J_regressor)[-1]
    seq = run_clips(model, feats, inits, T, J_regressor=J_regressor, keep=('kp_3d', 'verts', 'rotmat', 'theta') if avg_filter else ('kp_3d', 'verts'))
    recs = []
    for s, i in enumerate(mine):
        pred_j3d = torch.cat([boot['kp_3d'][s, :T - 1], seq[s]['kp_3d']], dim=0)
        pred_verts = torch.cat([boot['verts'][s, :T - 1], seq[s]['verts']], dim=0)
        if avg_filter:                                                     # evaluate.py:273-291
            from .filters import smooth_pose_mat
            rot = torch.cat([boot['rotmat'][s, :T - 1].reshape(-1, 24, 3, 3), seq[s]['rotmat'].reshape(-1, 24, 3, 3)], dim=0).contiguous()
            betas = torch.cat([boot['theta'][s, :T - 1, 75:], seq[s]['theta'][:, 75:]], dim=0).contiguous()
            rot = smooth_pose_mat(rot, ratio=0.3)
            mesh = model.regressor.smpl(betas=betas, body_pose=rot[:, 1:], global_orient=rot[:, 0:1], pose2rot=False).vertices
            pred_j3d = model._engine.joints_from_verts(mesh.contiguous(), J_regressor)
        rec = clip_metric_record(model, i, clips[names[i]], pred_j3d, pred_verts, dataset)
        if rec is not None:
            recs.append(rec)
    out = torch.stack(recs) if recs else torch.zeros(0, 8, dtype=torch.float64, device=dev)
    return out, mine


@torch.no_grad()
def measure_step_ms(model, seqlen, batches=(1, 2, 5, 10, 19, 37), steps=48, J_regressor=None):
    """{active clips: ms per lock-step of run_clips} measured on this GPU: B synthetic clips of equal length advance `steps` windows in lock-step (what
    distributed.StepCost interpolates; the committed default table came from this function)."""
    import time
    from . import synth
    dev = next(model.parameters()).device
    T = int(seqlen)
    out = {}
    for B in batches:
        w = torch.from_numpy(synth.synthetic_windows(int(B), T - 1 + steps, 4242)).to(dev)
        feats = [w[b, :, :2048].contiguous() for b in range(int(B))]
        inits = [w[b, :T - 1, 2048:].contiguous() for b in range(int(B))]
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_clips(model, feats, inits, T, J_regressor=J_regressor, keep=('kp_3d', 'verts'))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps * 1e3
            if rep and (best is None or dt < best):
                best = dt
        out[int(B)] = best
    return out


def gather_and_reduce(records):
    """All ranks call; rank 0 gets the metric dict (evaluate.py:459-462), others None."""
    allr = D.gather_records(records, dst=0)
    if allr is None:
        return None
    return M.reduce_records(allr) if allr.shape[0] else {}


def gather_rank_stats(rank, seconds, lengths, mine):
    """Per-rank load report on rank 0: wall seconds, clips, frames and the longest clip of every rank.
    Windows of a clip are serial (evaluate.py:247-269), so a rank's time is set by its longest clip's chain of
    lock-step window steps, not by its frame total; `seconds_max_over_mean` is the measured imbalance."""
    row = torch.tensor([[float(rank), float(seconds), float(len(mine)), float(sum(lengths[i] for i in mine)),
                         float(max([lengths[i] for i in mine], default=0))]], dtype=torch.float64)
    if D.dist.is_available() and D.dist.is_initialized():
        row = row.to(torch.device('cuda', torch.cuda.current_device())) if D.dist.get_backend() == 'nccl' else row
    allr = D.gather_records(row, dst=0)
    if allr is None:
        return None
    allr = allr.cpu()
    allr = allr[allr[:, 0].argsort()]
    secs = allr[:, 1]
    return {'seconds': [float(v) for v in secs], 'clips': [int(v) for v in allr[:, 2]],
            'frames': [int(v) for v in allr[:, 3]], 'longest_clip_frames': [int(v) for v in allr[:, 4]],
            'seconds_max_over_mean': float(secs.max() / secs.mean()) if float(secs.mean()) > 0 else 1.0}
