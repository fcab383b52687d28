"""Drop-in for the reference's lib/models/smpl.py: `SMPL`, joint tables and data paths.

The reference class subclasses `smplx.SMPL` (third-party, not installed here); this one
is a plain nn.Module that owns the same tables as buffers (same state-dict key names as
smplx registers them, so checkpoints that contain `regressor.smpl.*` load) and hands them
to the HIP library.  It computes nothing itself: the forward of the model it belongs to
runs LBS in libtepose_hip.so (tepose_amd/csrc/smpl.hip).
"""
import os
import os.path as osp
import pickle
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

BASE_DATA_DIR = os.environ.get('TEPOSE_BASE_DATA_DIR', 'data/base_data')   # lib/core/config.py

# Map joints to SMPL joints (values of reference lib/models/smpl.py:14-51)
JOINT_MAP = {
    'OP Nose': 24, 'OP Neck': 12, 'OP RShoulder': 17, 'OP RElbow': 19, 'OP RWrist': 21,
    'OP LShoulder': 16, 'OP LElbow': 18, 'OP LWrist': 20, 'OP MidHip': 0, 'OP RHip': 2,
    'OP RKnee': 5, 'OP RAnkle': 8, 'OP LHip': 1, 'OP LKnee': 4, 'OP LAnkle': 7, 'OP REye': 25,
    'OP LEye': 26, 'OP REar': 27, 'OP LEar': 28, 'OP LBigToe': 29, 'OP LSmallToe': 30,
    'OP LHeel': 31, 'OP RBigToe': 32, 'OP RSmallToe': 33, 'OP RHeel': 34, 'Right Ankle': 8,
    'Right Knee': 5, 'Right Hip': 45, 'Left Hip': 46, 'Left Knee': 4, 'Left Ankle': 7,
    'Right Wrist': 21, 'Right Elbow': 19, 'Right Shoulder': 17, 'Left Shoulder': 16,
    'Left Elbow': 18, 'Left Wrist': 20, 'Neck (LSP)': 47, 'Top of Head (LSP)': 48,
    'Pelvis (MPII)': 49, 'Thorax (MPII)': 50, 'Spine (H36M)': 51, 'Jaw (H36M)': 52,
    'Head (H36M)': 53, 'Nose': 24, 'Left Eye': 26, 'Right Eye': 25, 'Left Ear': 28, 'Right Ear': 27,
}
JOINT_NAMES = [
    'OP Nose', 'OP Neck', 'OP RShoulder', 'OP RElbow', 'OP RWrist', 'OP LShoulder', 'OP LElbow',
    'OP LWrist', 'OP MidHip', 'OP RHip', 'OP RKnee', 'OP RAnkle', 'OP LHip', 'OP LKnee',
    'OP LAnkle', 'OP REye', 'OP LEye', 'OP REar', 'OP LEar', 'OP LBigToe', 'OP LSmallToe',
    'OP LHeel', 'OP RBigToe', 'OP RSmallToe', 'OP RHeel', 'Right Ankle', 'Right Knee', 'Right Hip',
    'Left Hip', 'Left Knee', 'Left Ankle', 'Right Wrist', 'Right Elbow', 'Right Shoulder',
    'Left Shoulder', 'Left Elbow', 'Left Wrist', 'Neck (LSP)', 'Top of Head (LSP)', 'Pelvis (MPII)',
    'Thorax (MPII)', 'Spine (H36M)', 'Jaw (H36M)', 'Head (H36M)', 'Nose', 'Left Eye', 'Right Eye',
    'Left Ear', 'Right Ear',
]
JOINT_IDS = {JOINT_NAMES[i]: i for i in range(len(JOINT_NAMES))}
JOINT_REGRESSOR_TRAIN_EXTRA = osp.join(BASE_DATA_DIR, 'J_regressor_extra.npy')
SMPL_MEAN_PARAMS = osp.join(BASE_DATA_DIR, 'smpl_mean_params.npz')
SMPL_MODEL_DIR = BASE_DATA_DIR
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]

_TABLES = ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights')

SMPLOutput = namedtuple('SMPLOutput', 'vertices global_orient body_pose joints betas full_pose')


class _ChStub(object):
    """Stand-in for chumpy objects inside the official SMPL pickles (chumpy is not needed to
    read the arrays: a Ch object's state carries its value under 'x')."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'x': state})


class _SmplUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split('.')[0] == 'chumpy':
            return _ChStub
        return super().find_class(module, name)


def _arr(v):
    if isinstance(v, _ChStub):
        v = v.__dict__.get('x', v.__dict__.get('r'))
    if hasattr(v, 'toarray'):
        v = v.toarray()
    return np.asarray(v)


def load_smpl_pkl(path):
    """SMPL_{NEUTRAL,MALE,FEMALE}.pkl -> dict of float32 arrays in smplx's buffer layout
    (posedirs [207, 20670], shapedirs[..., :10]).  Untested against a real file here (the
    models are licence-gated)."""
    with open(path, 'rb') as f:
        d = _SmplUnpickler(f, encoding='latin1').load()
    posedirs = _arr(d['posedirs']).astype(np.float32)            # [6890,3,207]
    return {
        'v_template': _arr(d['v_template']).astype(np.float32),
        'shapedirs': _arr(d['shapedirs']).astype(np.float32)[:, :, :10],
        'posedirs': posedirs.reshape(-1, posedirs.shape[-1]).T.copy(),
        'J_regressor': _arr(d['J_regressor']).astype(np.float32),
        'lbs_weights': _arr(d['weights']).astype(np.float32),
        'parents': _arr(d['kintree_table'])[0].astype(np.int64),
        'faces': _arr(d['f']).astype(np.int64),
    }


class _VertexJointSelector(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('extra_joints_idxs', torch.tensor(
            [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
             2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133], dtype=torch.long))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # smplx-owned constant: a checkpoint may or may not carry it (the author's smplx version decides); the table above is
        # what the hot path uses either way -- never "missing", never a shape error (evaluate.py:124 loads strictly)
        state_dict.pop(prefix + 'extra_joints_idxs', None)


class SMPL(nn.Module):
    """SMPL(model_path, batch_size=1, create_transl=False, gender='neutral', ...) as the callers
    build it (lib/models/spin.py:226-230, evaluate.py:130-135, demo.py:150-155)."""

    def __init__(self, model_path=SMPL_MODEL_DIR, batch_size=1, create_transl=False, gender='neutral',
                 tables=None, **kwargs):
        super().__init__()
        if tables is None:
            path = model_path
            if osp.isdir(path):
                path = osp.join(path, 'SMPL_%s.pkl' % gender.upper())
            if not osp.isfile(path):
                raise FileNotFoundError('SMPL model file %r does not exist (licence-gated download, '
                                        'reference README.md:21-25)' % path)
            tables = load_smpl_pkl(path)
            tables['J_regressor_extra'] = np.load(JOINT_REGRESSOR_TRAIN_EXTRA)
        self.batch_size = batch_size
        self.gender = gender
        for k in _TABLES:
            self.register_buffer(k, torch.as_tensor(np.asarray(tables[k]), dtype=torch.float32))
        par = torch.as_tensor(np.asarray(tables['parents']), dtype=torch.long).clone()
        par[0] = -1
        self.register_buffer('parents', par)
        faces = tables.get('faces')
        self.faces = None if faces is None else np.asarray(faces)
        if faces is not None:
            self.register_buffer('faces_tensor', torch.as_tensor(np.asarray(faces), dtype=torch.long))
        # parameters smplx creates (never read by the hot path; kept for state-dict parity)
        self.betas = nn.Parameter(torch.zeros(batch_size, 10))
        self.global_orient = nn.Parameter(torch.zeros(batch_size, 3))
        self.body_pose = nn.Parameter(torch.zeros(batch_size, 69))
        self.vertex_joint_selector = _VertexJointSelector()
        self.register_buffer('J_regressor_extra',
                             torch.as_tensor(np.asarray(tables['J_regressor_extra']), dtype=torch.float32))
        self.joint_map = torch.tensor([JOINT_MAP[i] for i in JOINT_NAMES], dtype=torch.long)

    @classmethod
    def from_tables(cls, tables, batch_size=64, gender='neutral'):
        """Build from in-memory arrays (synthetic tables in tests/bench)."""
        return cls(model_path=None, batch_size=batch_size, gender=gender, tables=tables)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        # Checkpoints carry whatever the author's smplx version registered under
        # `regressor.smpl.*` (evaluate.py:124 loads strictly): take entries whose name and
        # shape match, ignore the rest, never report this module's keys as missing.
        own = dict(self.named_parameters(recurse=False))
        own.update(dict(self.named_buffers(recurse=False)))
        for k in [k for k in state_dict if k.startswith(prefix)]:
            name = k[len(prefix):]
            if '.' in name:
                continue            # child modules handle themselves
            v = state_dict[k]
            if name in own and tuple(own[name].shape) == tuple(v.shape):
                with torch.no_grad():
                    own[name].copy_(v)
            else:
                state_dict.pop(k)   # unknown smplx-owned key: accepted and ignored

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True, **kwargs):
        """Standalone call as the reference makes it (lib/utils/smooth_pose.py:35-64,
        evaluate.py:279-286, lib/utils/eval_utils.py:155-169): returns an SMPLOutput with
        .vertices [N,6890,3] and the wrapper's 49 .joints (lib/models/smpl.py:72-84).
        LBS runs in libtepose_hip.so on the module's cuda device (or the current one when the
        module and inputs live on the CPU; results come back on the inputs' device)."""
        from .engine import Engine
        ref = next(t for t in (betas, body_pose, global_orient) if t is not None)
        out_dev = ref.device
        dev = self.v_template.device if self.v_template.is_cuda else (
            out_dev if out_dev.type == 'cuda' else torch.device('cuda', torch.cuda.current_device()))
        n = ref.shape[0]

        def prep(t, default):
            t = default[:1].expand(n, -1) if t is None else t
            return t.detach().to(dev, torch.float32)
        betas_d = prep(betas, self.betas).reshape(n, 10).contiguous()
        if pose2rot:
            full = torch.cat([prep(global_orient, self.global_orient).reshape(n, 3),
                              prep(body_pose, self.body_pose).reshape(n, 69)], dim=1).contiguous()
        else:
            full = torch.cat([global_orient.detach().to(dev, torch.float32).reshape(n, 1, 3, 3),
                              body_pose.detach().to(dev, torch.float32).reshape(n, 23, 3, 3)], dim=1).contiguous()
        eng = self.__dict__.get('_engine')
        if eng is None:
            eng = Engine(1, 64)
            object.__setattr__(self, '_engine', eng)
        with torch.cuda.device(dev):
            eng.pack_smpl(self, dev)
            verts, joints = eng.smpl_fwd(full, betas_d, bool(pose2rot))
        return SMPLOutput(vertices=verts.to(out_dev), global_orient=global_orient, body_pose=body_pose,
                          joints=joints.to(out_dev), betas=betas, full_pose=full.to(out_dev))


def get_smpl_faces():
    return SMPL(SMPL_MODEL_DIR, batch_size=1, create_transl=False).faces
