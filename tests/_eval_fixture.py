"""Shared by the CPU and GPU tests of BASELINE config 4's evaluation branches: rebuild the synthetic `*_db.pt` a
tests/golden/eval_*.npz fixture was generated from (tests/golden/make_golden.py::eval_case -- the statements of the
reference's own evaluate.py, executed from the file as AST slices, with its TePose / VIBE classes and metric functions) and return it with the expected values."""
import os

import numpy as np

from tepose_amd import synth
from tepose_amd.data import split_db_into_clips, synthetic_eval_db

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = {'eval_mpii3d_L1H64_T5': 'mpii3d', 'eval_h36m_L1H64_T5': 'h36m', 'eval_h36m14_L1H64_T4': 'h36m',
         'eval_3dpw_L2H64_T6': '3dpw',
         # evaluate.py --filter (lines 273-291): the evaluated joints are those of the mesh re-posed with slerp-smoothed rotations
         'eval_3dpw_filter_L2H64_T6': '3dpw', 'eval_h36m14_filter_L1H64_T4': 'h36m'}


def load(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    L, H, T, seed_w, seed_db, joints = [int(v) for v in g['meta'][:6]]
    lens = [int(v) for v in g['meta'][6:]]
    db, pse = synthetic_eval_db(lens, seed=seed_db, joints=joints)
    for f in g['invalid_frames']:
        db['valid'][int(f)] = 0
    dataset = CASES[name]
    if dataset == 'mpii3d':
        db['valid_i'] = g['valid_i']
    clips = split_db_into_clips(db, pse, mpii3d=dataset == 'mpii3d')
    final = dict(zip([str(k) for k in g['final_keys']], [float(v) for v in g['final_values']]))
    per_clip = {}
    for ci in g['evaluated_clips']:
        ci = int(ci)
        per_clip[ci] = {k: g['clip%d_%s' % (ci, k)] for k in
                        ('raw_pred', 'mpjpe_all', 'pa_all', 'accel_all', 'pose_map', 'accel_map', 'has_accel', 'mpvpe')}
    return {'L': L, 'H': H, 'T': T, 'seed_w': seed_w, 'dataset': dataset, 'clips': clips, 'final': final,
            'avg_filter': bool(int(g['avg_filter'])) if 'avg_filter' in g.files else False,
            'per_clip': per_clip, 'tot_num_pose': int(g['tot_num_pose']), 'joints': joints}


def vibe_state(L, H, seed):
    vstate = synth.synthetic_vibe_state_dict(L, H, seed)
    mean = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0],
            'cam': vstate['regressor.init_cam'][0]}
    return vstate, mean


def write_base_data(base_dir, smpl_np, mean):
    """Fabricate the four `data/base_data` files in the layouts the reference reads (SMPL pickle in the official file
    layout: posedirs [V,3,207], 300 shape directions, sparse J_regressor, uint32 kintree_table)."""
    import pickle
    import scipy.sparse as sp
    os.makedirs(str(base_dir), exist_ok=True)
    np.save(os.path.join(str(base_dir), 'J_regressor_h36m.npy'), smpl_np['J_regressor_h36m'])
    np.save(os.path.join(str(base_dir), 'J_regressor_extra.npy'), smpl_np['J_regressor_extra'])
    np.savez(os.path.join(str(base_dir), 'smpl_mean_params.npz'), pose=np.asarray(mean['pose'], np.float32),
             shape=np.asarray(mean['shape'], np.float64), cam=np.asarray(mean['cam'], np.float32))
    par = np.asarray(smpl_np['parents']).astype(np.int64)
    kintree = np.stack([np.where(par < 0, 2 ** 32 - 1, par), np.arange(24)]).astype(np.uint32)
    d = {'v_template': smpl_np['v_template'].astype(np.float64),
         'shapedirs': np.concatenate([smpl_np['shapedirs'], np.zeros((6890, 3, 290), np.float32)], axis=2),
         'posedirs': smpl_np['posedirs'].T.reshape(6890, 3, 207).astype(np.float64),
         'J_regressor': sp.csc_matrix(smpl_np['J_regressor'].astype(np.float64)),
         'weights': smpl_np['lbs_weights'].astype(np.float64), 'kintree_table': kintree,
         'f': np.zeros((13776, 3), np.uint32)}
    with open(os.path.join(str(base_dir), 'SMPL_NEUTRAL.pkl'), 'wb') as f:
        pickle.dump(d, f, protocol=2)


def write_checkpoint(path, state_np, prefix=''):
    """A checkpoint file as lib/core/trainer.py:393-404 writes it (gen_state_dict + a numpy `performance`)."""
    import torch
    torch.save({'epoch': 3, 'gen_state_dict': {prefix + k: torch.from_numpy(np.asarray(v)) for k, v in state_np.items()},
                'performance': np.float64(51.2)}, str(path))


def write_cfg(path, title, layers, hidden, pretrained='', seqlen=6):
    """An experiment YAML with the key structure of the reference's configs/repr_*.yaml."""
    with open(str(path), 'w') as f:
        f.write("TITLE: '%s'\nDEVICE: 'cuda'\nDATASET:\n  SEQLEN: %d\n  VIDLEN: 520\nLOSS:\n  KP_2D_W: 300.0\n"
                "TRAIN:\n  BATCH_SIZE: 32\n  PRETRAINED: '%s'\n  PRETRAINED_REGRESSOR: 'data/base_data/spin_model_checkpoint.pth.tar'\n"
                "  DATASETS_3D:\n    - 'ThreeDPW'\n  MOT_DISCR:\n    OPTIM: 'Adam'\n    GCN:\n      num_class: 2\n"
                "MODEL:\n  TEMPORAL_TYPE: 'gru'\n  TGRU:\n    NUM_LAYERS: %d\n    HIDDEN_SIZE: %d\n"
                % (title, seqlen, pretrained, layers, hidden))
