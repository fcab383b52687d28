"""Evaluation metrics on the GPU and the per-clip records that ranks gather.

Mirrors evaluate.py:394-461: joint-format conversion, pelvis alignment, MPJPE, PA-MPJPE,
acceleration error, MPVPE (GT mesh = SMPL(theta_gt) with pose2rot=True), frame-weighted means.
Arithmetic runs in libtepose_hip.so (tepose_amd/csrc/metrics.hip); this file is plumbing."""
import torch

from . import _lib

# convert_kps(src='spin', dst=...) index tables (lib/data_utils/_kp_utils.py:28-38 copies by joint
# name; SURVEY.md 8f-2 lists the resulting indices; checked against the reference in make_golden.py)
SPIN_TO_COMMON = list(range(25, 39))
SPIN_TO_MPII3D_TEST = [38, 37, 33, 32, 31, 34, 35, 36, 27, 26, 25, 28, 29, 30, 39, 41, 43]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def joint_metrics(pred_j3d, target_j3d, pelvis='lsp'):
    """pred/target [N,J,3] cuda tensors (metres, same joint order) -> dict of [N] tensors in mm:
    'mpjpe', 'pa_mpjpe', 'accel' (accel[0] = accel[-1] = 0, as evaluate.py:439-440)."""
    lib = _lib.load()
    p = pred_j3d.float().contiguous()
    t = target_j3d.float().contiguous().to(p.device)
    N, J = p.shape[:2]
    out = {k: torch.empty(N, dtype=torch.float32, device=p.device) for k in ('mpjpe', 'pa_mpjpe', 'accel')}
    with torch.cuda.device(p.device):
        _lib.check(lib.tepose_metrics_joints(p.data_ptr(), t.data_ptr(), N, J, 0 if pelvis == 'lsp' else 1,
                                             out['mpjpe'].data_ptr(), out['pa_mpjpe'].data_ptr(),
                                             out['accel'].data_ptr(), _stream()), 'tepose_metrics_joints')
    return out


def gt_vertices(model, target_theta):
    """SMPL(betas=theta[:,75:], pose=theta[:,3:75], pose2rot=True).vertices with the model's
    current SMPL tables (lib/utils/eval_utils.py:155-169)."""
    eng = model._engine
    th = target_theta.float().contiguous()
    N = th.shape[0]
    with torch.cuda.device(th.device):
        eng.pack_regressor(model.regressor, th.device)
        ws = eng.workspace(max(1, (N + 1) // 2), 1, th.device)
        verts = torch.empty((N, 6890, 3), dtype=torch.float32, device=th.device)
        _lib.check(eng.lib.tepose_smpl_verts_from_theta(eng.handle, th.data_ptr(), N, verts.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), _stream()),
                   'tepose_smpl_verts_from_theta')
    return verts


def vertex_metric(pred_verts, target_verts):
    """MPVPE per frame in mm (lib/utils/eval_utils.py:173-175)."""
    lib = _lib.load()
    p, t = pred_verts.float().contiguous(), target_verts.float().contiguous()
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(lib.tepose_metrics_verts(p.data_ptr(), t.data_ptr(), p.shape[0], out.data_ptr(), _stream()),
                   'tepose_metrics_verts')
    return out


def clip_record(clip_id, m, mpvpe=None, valid_map=None):
    """One fixed-size row per clip for the rank-0 gather:
    [clip_id, n_pose, sum_mpjpe, sum_pa, n_accel, sum_accel, n_mpvpe, sum_mpvpe] (float64).
    valid_map: indices of evaluated frames (default all); the acceleration error drops the
    first and last frame of the clip (evaluate.py:441-450)."""
    n = m['mpjpe'].shape[0]
    dev = m['mpjpe'].device
    vm = torch.arange(n, device=dev) if valid_map is None else torch.as_tensor(valid_map, device=dev)
    rec = torch.zeros(8, dtype=torch.float64, device=dev)
    rec[0], rec[1] = float(clip_id), float(vm.numel())
    rec[2] = m['mpjpe'][vm].double().sum()
    rec[3] = m['pa_mpjpe'][vm].double().sum()
    if vm.numel() > 1:
        va = vm
        if int(va[0]) == 0:
            va = va[1:]
        if int(va[-1]) == n - 1:
            va = va[:-1]
        rec[4] = float(va.numel())
        rec[5] = m['accel'][va].double().sum()
    if mpvpe is not None:
        rec[6], rec[7] = float(mpvpe.numel()), mpvpe.double().sum()
    return rec


def reduce_records(records):
    """Frame-weighted means over all clips, exactly like np.mean(np.concatenate(...))
    in evaluate.py:461.  records: [n_clips, 8]."""
    r = records.double()
    out = {'mpjpe': float(r[:, 2].sum() / r[:, 1].sum()), 'mpjpe_pa': float(r[:, 3].sum() / r[:, 1].sum())}
    if r[:, 4].sum() > 0:
        out['accel_err'] = float(r[:, 5].sum() / r[:, 4].sum())
    if r[:, 6].sum() > 0:
        out['mpvpe'] = float(r[:, 7].sum() / r[:, 6].sum())
    return out


@torch.no_grad()
def trainer_evaluate(model, acc, target, seqlen):
    """Trainer.evaluate (lib/core/trainer.py:437-488) on the accumulators of one validation pass: acc = the dict
    tepose_amd.driver.validate_padded returns ('pred_kp_3d', 'pred_verts', 'pred_j3d_tsr'), target = the padded batch
    ('kp_3d' [C, vidlen, J, 3], 'theta' [C, vidlen, 85], 'vidlen_each' [C, 1]).  Returns the trainer's eval_dict in mm:
    'mpjpe', 'pa-mpjpe' (pelvis = mean of joints 2, 3; similarity Procrustes: the device kernels of joint_metrics),
    'accel', 'accel_err' (eval_utils.py:53-107: per video the second differences of frames seqlen-1 .. vidlen-3 resp.
    .. vidlen-5 of the [C, vidlen] tensors -- prediction pelvis-aligned per frame, target as trainer.py:470 leaves it --, normalised by sum(vidlen) - C (seqlen + 1) resp. (seqlen + 3)),
    'pve' (vertices against SMPL(target theta), eval_utils.py:141-175)."""
    T = int(seqlen)
    dev = acc['pred_kp_3d'].device
    lens = [int(v) for v in target['vidlen_each'].reshape(-1).tolist()]
    C = len(lens)
    nwin = [max(n - T + 1, 0) for n in lens]
    order = [(j, c) for j in range(max(nwin + [0])) for c in range(C) if j < nwin[c]]
    kp_t = target['kp_3d'].to(dev)
    tgt = torch.stack([kp_t[c, j + T - 1] for j, c in order])
    m = joint_metrics(acc['pred_kp_3d'].float().contiguous(), tgt.float().contiguous())
    out = {'mpjpe': float(m['mpjpe'].mean()), 'pa-mpjpe': float(m['pa_mpjpe'].mean())}
    p = acc['pred_j3d_tsr'].to(dev).clone()
    g = kp_t.clone().float()
    p -= (p[:, :, [2]] + p[:, :, [3]]) / 2.0
    # trainer.py:470 indexes the [C, vidlen, J, 3] target with `[:,[2],:]` / `[:,[3],:]`: FRAMES 2 and 3, not joints -- the
    # ground truth loses one constant per clip and joint and keeps its raw accelerations (pelvis motion included), while
    # the prediction (line above, trainer.py:469) is pelvis-aligned per frame.  Reproduced as written: accel_err of the
    # reference contains the ground-truth pelvis acceleration.
    g -= (g[:, [2]] + g[:, [3]]) / 2.0
    vl = torch.tensor(lens, dtype=torch.float32)
    a_p = p[:, 2:] - 2 * p[:, 1:-1] + p[:, :-2]
    a_g = g[:, 2:] - 2 * g[:, 1:-1] + g[:, :-2]
    an = a_p.norm(dim=3).mean(dim=2)                     # [C, vidlen - 2]
    en = (a_p - a_g).norm(dim=3).mean(dim=2)
    acc_sum = sum(float(an[c, T - 1:max(lens[c] - 2, T - 1)].sum()) for c in range(C))
    err_sum = sum(float(en[c, T - 1:max(lens[c] - 4, T - 1)].sum()) for c in range(C))
    out['accel'] = acc_sum / (float(vl.sum()) - C * (T + 1) + 1e-8) * 1000.0
    out['accel_err'] = err_sum / (float(vl.sum()) - C * (T + 3) + 1e-8) * 1000.0
    if 'pred_verts' in acc and acc['pred_verts'] is not None and 'theta' in target:
        th_t = target['theta'].to(dev)
        tt = torch.stack([th_t[c, j + T - 1] for j, c in order]).float().contiguous()
        out['pve'] = float(vertex_metric(acc['pred_verts'].float().contiguous(), gt_vertices(model, tt)).mean())
    return out
