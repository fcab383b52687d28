"""Shared by the CPU and GPU tests of BASELINE config 4's evaluation branches: rebuild the synthetic `*_db.pt` a
tests/golden/eval_*.npz fixture was generated from (tests/golden/make_golden.py::eval_case -- the reference's own
evaluate.py flow with its TePose / VIBE classes and metric functions) and return it with the expected values."""
import os

import numpy as np

from tepose_amd import synth
from tepose_amd.data import split_db_into_clips, synthetic_eval_db

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = {'eval_mpii3d_L1H64_T5': 'mpii3d', 'eval_h36m_L1H64_T5': 'h36m', 'eval_h36m14_L1H64_T4': 'h36m',
         'eval_3dpw_L2H64_T6': '3dpw'}


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
            'per_clip': per_clip, 'tot_num_pose': int(g['tot_num_pose']), 'joints': joints}


def vibe_state(L, H, seed):
    vstate = synth.synthetic_vibe_state_dict(L, H, seed)
    mean = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0],
            'cam': vstate['regressor.init_cam'][0]}
    return vstate, mean
