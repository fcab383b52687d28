"""Drop-in for the hot-path part of the reference's lib/models/spin.py: `Regressor`
(lib/models/spin.py:209-291) and `projection` (:307-351).  The ResNet-50 `HMR` backbone of
that file is out of scope (features are pre-extracted).
"""
import numpy as np
import torch
import torch.nn as nn

from .engine import Engine, on_device, warn_if_training
from .smpl import SMPL, SMPL_MEAN_PARAMS, SMPL_MODEL_DIR, H36M_TO_J14  # noqa: F401


class Regressor(nn.Module):
    """Same parameters, buffers and forward signature as the reference class; the forward
    (3x fc1/fc2/decoders, rot6d->R, SMPL LBS, joints, projection, R->axis-angle) is one call
    into libtepose_hip.so.

    Extra keyword `smpl=` takes a ready `SMPL` (the licence-gated model files are absent in
    this repo); `smpl_mean_params` may also be a dict with 'pose','shape','cam'."""

    def __init__(self, smpl_mean_params=SMPL_MEAN_PARAMS, smpl=None, _engine=None):
        super().__init__()
        npose = 24 * 6
        self.fc1 = nn.Linear(512 * 4 + npose + 13, 1024)
        self.drop1 = nn.Dropout()
        self.fc2 = nn.Linear(1024, 1024)
        self.drop2 = nn.Dropout()
        self.decpose = nn.Linear(1024, npose)
        self.decshape = nn.Linear(1024, 10)
        self.deccam = nn.Linear(1024, 3)
        nn.init.xavier_uniform_(self.decpose.weight, gain=0.01)
        nn.init.xavier_uniform_(self.decshape.weight, gain=0.01)
        nn.init.xavier_uniform_(self.deccam.weight, gain=0.01)
        self.smpl = smpl if smpl is not None else SMPL(SMPL_MODEL_DIR, batch_size=64, create_transl=False)
        mean_params = smpl_mean_params if isinstance(smpl_mean_params, dict) else np.load(smpl_mean_params)
        init_pose = torch.from_numpy(np.asarray(mean_params['pose'][:], dtype=np.float32)).unsqueeze(0)
        init_shape = torch.from_numpy(np.asarray(mean_params['shape'][:]).astype('float32')).unsqueeze(0)
        init_cam = torch.from_numpy(np.asarray(mean_params['cam'], dtype=np.float32)).unsqueeze(0)
        self.register_buffer('init_pose', init_pose)
        self.register_buffer('init_shape', init_shape)
        self.register_buffer('init_cam', init_cam)
        object.__setattr__(self, '_engine', _engine if _engine is not None else Engine(1, 64))

    def forward(self, x, init_pose=None, init_shape=None, init_cam=None, n_iter=3, is_train=False,
                J_regressor=None):
        warn_if_training(self, x)
        if not x.is_cuda:
            raise RuntimeError('tepose_amd runs on MI355X only: move the model and input to a cuda device')
        x = x.float().contiguous()
        eng = self._engine
        with on_device(x.device):
            eng.pack_regressor(self, x.device)
            use_j = J_regressor if (not is_train and J_regressor is not None) else None
            return [eng.regressor_fwd(x, n_iter, use_j, init=(init_pose, init_shape, init_cam))]


def warm_start_from_spin(regressor, ckpt_path):
    """Initialise the regressor from a SPIN checkpoint's 'model' entry, non-strictly, when the file
    exists -- what TePose.__init__ / VIBE.__init__ do with `pretrained`
    (lib/models/tepose.py:115-118, lib/models/vibe.py:97-101)."""
    import os
    if not ckpt_path or not os.path.isfile(ckpt_path):
        return False
    from .data import load_checkpoint
    weights = load_checkpoint(ckpt_path)['model']
    regressor.load_state_dict(weights, strict=False)
    print("=> loaded pretrained model from '%s'" % ckpt_path)
    return True


def projection(pred_joints, pred_camera):
    """lib/models/spin.py:307-320 with R = I and zero camera centre (host-side helper for
    callers; the model forward computes kp_2d on the GPU)."""
    t = torch.stack([pred_camera[:, 1], pred_camera[:, 2],
                     2 * 5000. / (224. * pred_camera[:, 0] + 1e-9)], dim=-1)
    p = pred_joints + t.unsqueeze(1)
    p = p / p[:, :, -1].unsqueeze(-1)
    return (5000. * p[:, :, :-1]) / (224. / 2.)


def hmr(*args, **kwargs):
    raise NotImplementedError('The ResNet-50 HMR feature extractor (lib/models/spin.py:16-206) is outside '
                              'the accelerated hot path; features are pre-extracted.')
