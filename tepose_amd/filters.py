"""Temporal post-filters of the reference, on the GPU.

`smooth_pose` mirrors lib/utils/smooth_pose.py:24-67 (OneEuro filter on the axis-angle pose, then
SMPL on the filtered pose) and `smooth_pose_mat` mirrors evaluate.py:32-59 (quaternion slerp
smoothing of the rotation matrices).  The recursions run in libtepose_hip.so (csrc/filters.hip);
the SMPL re-run is one batched call instead of the reference's per-frame loop."""
import numpy as np
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _to_dev(a, device):
    t = torch.as_tensor(a)
    return t.detach().to(device=device, dtype=torch.float32).contiguous().clone()


def one_euro(x, min_cutoff=0.004, beta=0.7, d_cutoff=1.0, device=None):
    """x [N, ...] -> filtered copy (frame 0 unchanged)."""
    dev = torch.device(device) if device is not None else (x.device if torch.is_tensor(x) and x.is_cuda
                                                          else torch.device('cuda', torch.cuda.current_device()))
    t = _to_dev(x, dev)
    n = t.shape[0]
    d = t[0].numel()
    with torch.cuda.device(dev):
        _lib.check(_lib.load().tepose_filter_one_euro(t.data_ptr(), n, d, float(min_cutoff), float(beta),
                                                     float(d_cutoff), _stream()), 'tepose_filter_one_euro')
    return t


def smooth_pose_mat(pose, ratio=0.3, device=None):
    """pose [N, 24, 3, 3] rotation matrices -> slerp-smoothed [N, 24, 3, 3] (same container kind)."""
    is_np = isinstance(pose, np.ndarray)
    dev = torch.device(device) if device is not None else (pose.device if (not is_np and pose.is_cuda)
                                                          else torch.device('cuda', torch.cuda.current_device()))
    t = _to_dev(pose, dev)
    n, j = t.shape[:2]
    out = torch.empty_like(t)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().tepose_filter_slerp(t.data_ptr(), out.data_ptr(), n, j, float(ratio), _stream()),
                   'tepose_filter_slerp')
    return out.cpu().numpy() if is_np else out


def smooth_pose(pred_pose, pred_betas, smpl, min_cutoff=0.004, beta=0.7):
    """lib/utils/smooth_pose.py:24-67.  pred_pose [N,24,3] or [N,72] axis-angle, pred_betas [N,10];
    `smpl`: a tepose_amd.SMPL.  Returns (verts [N,6890,3], pose_hat [N,24,3], joints3d [N,49,3]) as
    numpy arrays when numpy came in, tensors otherwise."""
    is_np = isinstance(pred_pose, np.ndarray)
    n = pred_pose.shape[0]
    pose_hat = one_euro(pred_pose, min_cutoff=min_cutoff, beta=beta).reshape(n, 24, 3)
    betas = torch.as_tensor(pred_betas).to(pose_hat.device, torch.float32)
    out = smpl(betas=betas, body_pose=pose_hat[:, 1:], global_orient=pose_hat[:, 0:1])
    res = (out.vertices, pose_hat, out.joints)
    return tuple(r.cpu().numpy() for r in res) if is_np else res
