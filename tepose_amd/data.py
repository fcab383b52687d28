"""Readers for the reference's on-disk formats on the evaluation path.

No sample of these files exists in the reference repo (they are produced by its offline
preprocessing or are licence-gated downloads), so the layouts follow the code that writes and
reads them: `*_db.pt` = joblib dict of per-frame arrays (lib/data_utils/threedpw_utils.py:47-58,
147-159), `*_pseudotheta.pt` = joblib [N,85] array (lib/data_utils/pseudo_theta.py:102-105),
checkpoints = torch dict with 'gen_state_dict' (lib/core/trainer.py:393-404)."""
from collections import OrderedDict

import numpy as np


def split_db_into_clips(db, pseudotheta, target_action='', mpii3d=False):
    """evaluate.py:171-206: group frames by `vid_name` (np.unique order), keep `valid` frames,
    force the pseudo-theta camera to [1,0,0].  Returns OrderedDict name -> dict of arrays."""
    names = db['vid_name']
    pse = np.array(pseudotheta, dtype=np.float32, copy=True)
    pse[:, :3] = np.array([1., 0., 0.], dtype=np.float32)
    clips = OrderedDict()
    for u in np.unique(names):
        if target_action != '' and target_action not in u:
            continue
        idx = names == u
        valids = db['valid'][idx].astype(bool) if 'valid' in db else np.ones(int(idx.sum()), dtype=bool)
        c = {'features': db['features'][idx][valids], 'joints3D': db['joints3D'][idx][valids],
             'theta_pseu': pse[idx][valids]}
        if mpii3d:
            c['pose'] = np.zeros((len(valids), 72))
            c['shape'] = np.zeros((len(valids), 10))
            c['valid_i'] = db['valid_i'][idx][valids]
        else:
            c['pose'] = db['pose'][idx][valids]
            c['shape'] = db['shape'][idx][valids]
        clips[str(u)] = c
    return clips


def load_eval_db(db_path, pseudotheta_path, target_action=''):
    import joblib
    return split_db_into_clips(joblib.load(db_path), joblib.load(pseudotheta_path), target_action,
                               mpii3d='mpii3d' in str(db_path))


def load_base_data(base_dir, gender='neutral'):
    """The four files of `data/base_data` the evaluation reads (evaluate.py:109,130-135; lib/models/smpl.py:54-56,67;
    lib/models/spin.py:232-235): `J_regressor_h36m.npy` [17,6890], `smpl_mean_params.npz` (pose[144], shape[10], cam[3]),
    `SMPL_<GENDER>.pkl` (chumpy-free reader) and `J_regressor_extra.npy` [9,6890].  Returns
    {'J_regressor_h36m', 'mean_params', 'smpl_tables'} with float32 arrays, the tables in the layout
    tepose_amd.smpl.SMPL.from_tables takes.  Every missing file is named in one FileNotFoundError."""
    import os.path as osp
    from .smpl import load_smpl_pkl
    files = {'J_regressor_h36m': 'J_regressor_h36m.npy', 'mean_params': 'smpl_mean_params.npz',
             'smpl': 'SMPL_%s.pkl' % gender.upper(), 'J_regressor_extra': 'J_regressor_extra.npy'}
    missing = [f for f in files.values() if not osp.isfile(osp.join(str(base_dir), f))]
    if missing:
        raise FileNotFoundError('base data directory %r lacks %s (licence-gated downloads, reference README.md:21-25)'
                                % (str(base_dir), ', '.join(missing)))
    jh = np.asarray(np.load(osp.join(str(base_dir), files['J_regressor_h36m'])), dtype=np.float32)
    je = np.asarray(np.load(osp.join(str(base_dir), files['J_regressor_extra'])), dtype=np.float32)
    mp = np.load(osp.join(str(base_dir), files['mean_params']))
    mean = {'pose': np.asarray(mp['pose'], dtype=np.float32).reshape(-1), 'shape': np.asarray(mp['shape'], dtype=np.float32).reshape(-1),
            'cam': np.asarray(mp['cam'], dtype=np.float32).reshape(-1)}
    if jh.shape != (17, 6890) or je.shape != (9, 6890) or mean['pose'].shape != (144,) or mean['shape'].shape != (10,) \
            or mean['cam'].shape != (3,):
        raise ValueError('base data shapes: J_regressor_h36m %s (want 17x6890), J_regressor_extra %s (9x6890), mean pose %s '
                         '(144) shape %s (10) cam %s (3)' % (jh.shape, je.shape, mean['pose'].shape, mean['shape'].shape,
                                                              mean['cam'].shape))
    tables = load_smpl_pkl(osp.join(str(base_dir), files['smpl']))
    tables['J_regressor_extra'] = je
    return {'J_regressor_h36m': jh, 'mean_params': mean, 'smpl_tables': tables}


def load_generator_state_dict(path, map_location='cpu'):
    """checkpoint['gen_state_dict'] with a DataParallel 'module.' prefix stripped
    (evaluate.py:121-124, lib/utils/utils.py:40-45)."""
    ckpt = load_checkpoint(path, map_location)
    sd = ckpt['gen_state_dict'] if 'gen_state_dict' in ckpt else ckpt
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())


def load_checkpoint(path, map_location='cpu'):
    """torch.load for the reference's checkpoint files.  Its trainer stores `performance` as the np.float64 that
    evaluate() returns next to the state dicts, optimizer and lr_scheduler states (lib/core/trainer.py:393-404,413,503),
    so the weights-only unpickler needs numpy's scalar / dtype reconstructors allow-listed; nothing else is admitted
    (no weights_only=False fallback: a checkpoint is a download)."""
    import numpy as np
    import torch
    allowed = [np.dtype, np.float64, np.float32, np.int64, np.ndarray]
    try:
        from numpy._core import multiarray as _ma            # numpy >= 2
    except ImportError:                                      # numpy 1.x
        from numpy.core import multiarray as _ma
    allowed += [_ma.scalar, _ma._reconstruct]
    allowed += [type(np.dtype(t)) for t in ('float64', 'float32', 'int64', 'int32', 'uint8', 'bool')]
    with torch.serialization.safe_globals(allowed):
        return torch.load(path, map_location=map_location, weights_only=True)


def synthetic_eval_db(lengths, seed=0, joints=49):
    """Synthetic stand-in with the schema of a 3DPW `*_db.pt` + `*_pseudotheta.pt` pair."""
    from . import synth
    n = int(sum(lengths))
    names = np.concatenate([np.array(['clip_%02d' % i] * int(l)) for i, l in enumerate(lengths)])
    w = synth.synthetic_windows(1, n, 9000 + seed)[0]
    theta = synth.synthetic_windows(1, n + 1, 9100 + seed)[0, :n, 2048:]
    db = {'vid_name': names, 'features': w[:, :2048].copy(),
          'joints3D': synth.normal('db%d/j3d' % seed, (n, joints, 3), std=0.3),
          'pose': theta[:, 3:75].copy(), 'shape': theta[:, 75:].copy(),
          'valid': np.ones(n, dtype=np.float32)}
    pse = synth.synthetic_windows(1, n + 1, 9200 + seed)[0, :n, 2048:].copy()
    return db, pse


def padded_validation_batch(db, pseudotheta, seqlen, joints=14, eval_class=None):
    """The batch the reference's validation Datasets hand to trainer.validate (lib/dataset/threedpw_test.py:54-134,
    h36m_val.py; lib/data_utils/_img_utils.py:356-376), hot-path fields only.  A database whose `joints3D` hold the 49 'spin' joints (Human3.6M:
    h36m_val.py:77 `convert_kps(..., src='spin', dst='common')`) is converted to the 14 common joints first; a 14-joint database (3DPW test) is taken as is.
    eval_class (the config's TRAIN.DATASET_EVAL, lib/dataset/loaders.py:118): None / 'ThreeDPW' / 'ThreeDPW_TEST' / 'Human36M_VAL' = the above;
    'Human36M' (Dataset3D on H3.6M, dataset_3d.py:189-194,214-219): common joints as above, but ground-truth pose and shape ZERO; 'MPII3D'
    (dataset_3d.py:182-187,231-233): the 17 `mpii3d_test` joints, pose and shape zero.
    Videos in order of first appearance of their
    `vid_name`, those shorter than `seqlen` dropped, every clip zero-padded to the longest; the arrays are staged in
    float16 exactly as the Datasets do (`np.zeros(..., dtype=np.float16)` filled, then `.float()`), so features and
    thetas carry fp16 rounding; theta / theta_pseu = [1, 0, 0 | pose 72 | shape 10].
    Returns dict of torch tensors: features [C, vidlen, 2048], theta, theta_pseu [C, vidlen, 85], kp_3d [C, vidlen,
    joints, 3], vidlen_each [C, 1], index [C, 1] -- what tepose_amd.driver.validate_padded consumes."""
    import torch
    names = np.asarray(db['vid_name'])
    _, first = np.unique(names, return_index=True)
    starts = np.sort(first)
    bounds = list(starts) + [names.shape[0]]
    spans = [(int(bounds[i]), int(bounds[i + 1])) for i in range(len(starts)) if bounds[i + 1] - bounds[i] >= seqlen]
    if not spans:
        return None
    C, vidlen = len(spans), max(e - s for s, e in spans)
    if eval_class not in (None, 'ThreeDPW', 'ThreeDPW_TEST', 'Human36M_VAL', 'Human36M', 'MPII3D'):
        raise ValueError('unknown validation dataset class %r' % (eval_class,))
    zero_gt = eval_class in ('Human36M', 'MPII3D')
    if eval_class == 'MPII3D':
        joints = 17
    pse = np.asarray(pseudotheta, dtype=np.float32)
    feats = np.zeros((C, vidlen, 2048), dtype=np.float16)
    theta = np.zeros((C, vidlen, 85), dtype=np.float16)
    theta_pseu = np.zeros((C, vidlen, 85), dtype=np.float16)
    kp_3d = np.zeros((C, vidlen, joints, 3), dtype=np.float16)
    cam = np.tile(np.array([1., 0., 0.], dtype=np.float32), (vidlen, 1))
    for c, (s, e) in enumerate(spans):
        n = e - s
        feats[c, :n] = db['features'][s:e]
        theta[c, :n] = np.concatenate([cam[:n], np.zeros((n, 82), np.float32)], axis=1) if zero_gt else \
            np.concatenate([cam[:n], db['pose'][s:e], db['shape'][s:e]], axis=1)
        theta_pseu[c, :n] = np.concatenate([cam[:n], pse[s:e, 3:75], pse[s:e, 75:]], axis=1)
        j3 = np.asarray(db['joints3D'][s:e])
        if eval_class == 'MPII3D':                               # dataset_3d.py:187: spin -> mpii3d_test (17 joints)
            from .metrics import SPIN_TO_MPII3D_TEST
            j3 = j3[:, SPIN_TO_MPII3D_TEST]
        elif j3.shape[1] == 49:                                  # h36m_val.py:77 / dataset_3d.py:194: spin -> common (the 14 LSP joints), then [:nj]
            from .metrics import SPIN_TO_COMMON
            j3 = j3[:, SPIN_TO_COMMON]
        kp_3d[c, :n] = j3[:, :joints]
    return {'features': torch.from_numpy(feats).float(), 'theta': torch.from_numpy(theta).float(),
            'theta_pseu': torch.from_numpy(theta_pseu).float(), 'kp_3d': torch.from_numpy(kp_3d).float(),
            'vidlen_each': torch.tensor([float(e - s) for s, e in spans]).view(C, 1),
            'index': torch.arange(C).float().view(C, 1)}
