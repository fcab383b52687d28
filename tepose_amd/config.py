"""The reference's experiment configuration, as far as the per-window inference path reads it.

lib/core/config.py builds a yacs CfgNode of defaults and merges a YAML file over it (`update_cfg`, :129-132); evaluate.py /
demo.py then read MODEL.TGRU.{NUM_LAYERS, HIDDEN_SIZE}, DATASET.SEQLEN, TRAIN.{BATCH_SIZE, PRETRAINED,
PRETRAINED_REGRESSOR}, DEVICE and TITLE (evaluate.py:113-127,146-152).  yacs is not part of this image and none of its
machinery is needed to read six keys: PyYAML + the same defaults + the same merge rule (a key of the file replaces the
default, nested nodes merge, a key that the defaults do not have is an error -- yacs raises KeyError there too)."""
import copy
import os.path as osp


class CfgNode(dict):
    """Attribute access over nested dicts (cfg.MODEL.TGRU.NUM_LAYERS), like yacs' CfgNode."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)


def _node(d):
    return CfgNode({k: _node(v) if isinstance(v, dict) else v for k, v in d.items()})


# defaults of lib/core/config.py:33-126 -- the sections the evaluation / demo path reads, plus the names of the others so
# that the shipped YAML files (which set LOSS / TRAIN.MOT_DISCR ... too) merge without "unknown key" errors
_DEFAULTS = {
    'TITLE': 'default', 'OUTPUT_DIR': 'results', 'EXP_NAME': 'default', 'DEVICE': 'cuda', 'DEBUG': True, 'LOGDIR': '',
    'NUM_WORKERS': 8, 'DEBUG_FREQ': 1000, 'SEED_VALUE': -1, 'render': False,
    'CUDNN': {'BENCHMARK': True, 'DETERMINISTIC': False, 'ENABLED': True},
    'TRAIN': {'DATASETS_2D': ['Insta'], 'DATASETS_3D': ['MPII3D'], 'DATASET_EVAL': 'ThreeDPW', 'BATCH_SIZE': 32,
              'OVERLAP': True, 'DATA_2D_RATIO': 0.5, 'START_EPOCH': 0, 'END_EPOCH': 5, 'PRETRAINED_REGRESSOR': '',
              'PRETRAINED': '', 'RESUME': '', 'NUM_ITERS_PER_EPOCH': 1000, 'UPDATE_THETA_RATE': 1.0, 'LR_PATIENCE': 5,
              'GEN_OPTIM': 'Adam', 'GEN_LR': 1e-4, 'GEN_WD': 1e-4, 'GEN_MOMENTUM': 0.9,
              'MOT_DISCR': {'OPTIM': 'SGD', 'LR': 1e-2, 'WD': 1e-4, 'MOMENTUM': 0.9, 'NUM_CLASS': 2, 'UPDATE_STEPS': 1,
                            'FEATURE_POOL': 'concat', 'HIDDEN_SIZE': 1024, 'NUM_LAYERS': 1,
                            'GCN': {'num_class': 2, 'num_point': 24, 'num_person': 1, 'num_gcn_scales': 13,
                                    'num_g3d_scales': 6, 'graph': 'lib.graph.smplx_theta.AdjMatrixGraph'}}},
    'DATASET': {'SEQLEN': 20, 'VIDLEN': 1000, 'OVERLAP': 0.5},
    'LOSS': {'KP_2D_W': 60., 'KP_3D_W': 30., 'SHAPE_W': 0.001, 'POSE_W': 1.0, 'D_MOTION_LOSS_W': 1.},
    'MODEL': {'TEMPORAL_TYPE': 'gru', 'TGRU': {'NUM_LAYERS': 1, 'HIDDEN_SIZE': 2048}},
}


def get_cfg_defaults():
    return _node(_DEFAULTS)


def _merge(dst, src, path):
    for k, v in src.items():
        if k not in dst:
            raise KeyError('Non-existent config key: %s' % '.'.join(path + [k]))
        if isinstance(dst[k], dict):
            if not isinstance(v, dict):
                raise ValueError('config key %s is a section, the file gives %r' % ('.'.join(path + [k]), v))
            _merge(dst[k], v, path + [k])
        else:
            if dst[k] is not None and v is not None and type(v) is not type(dst[k]):
                if isinstance(dst[k], float) and isinstance(v, int) and not isinstance(v, bool):
                    v = float(v)                                     # yacs allows int -> float
                elif isinstance(dst[k], (list, tuple)) and isinstance(v, (list, tuple)):
                    v = type(dst[k])(v)
                else:
                    raise ValueError('Type mismatch for config key %s: %r vs default %r' % ('.'.join(path + [k]), v, dst[k]))
            dst[k] = v


def update_cfg(cfg_file):
    """lib/core/config.py:129-132: defaults merged with the YAML file."""
    import yaml
    cfg = get_cfg_defaults()
    with open(cfg_file) as f:
        loaded = yaml.safe_load(f) or {}
    _merge(cfg, loaded, [])
    return cfg


def model_kwargs(cfg, seqlen=None):
    """Constructor arguments of TePose as evaluate.py:113-119 passes them."""
    return {'n_layers': int(cfg.MODEL.TGRU.NUM_LAYERS), 'batch_size': int(cfg.TRAIN.BATCH_SIZE),
            'seqlen': int(cfg.DATASET.SEQLEN if seqlen is None else seqlen), 'hidden_size': int(cfg.MODEL.TGRU.HIDDEN_SIZE),
            'pretrained': cfg.TRAIN.PRETRAINED_REGRESSOR}


EVAL_SEQLEN = 6          # evaluate.py:141 hard-codes the evaluation window, whatever DATASET.SEQLEN says


def eval_db_paths(cfg, dataset, db_dir='data/preprocessed_data', render=False):
    """The database / pseudo-theta file names evaluate.py:146-162 derives from the dataset and the config's TITLE.
    Returns (db_path, pseudotheta_path); raises for a dataset / title combination the reference has no file for."""
    if dataset == '3dpw':
        stem = '3dpw_test%s' % ('_all' if render else '')
    elif dataset == 'h36m':
        if cfg.TITLE == 'repr_wpw_h36m_mpii3d_model':
            stem = 'h36m_test_25fps_nosmpl'                              # Table 1
        elif cfg.TITLE == 'repr_wopw_h36m_model':
            stem = 'h36m_test_front_25fps_tight_nosmpl'                  # Table 2
        else:
            raise ValueError('evaluate.py:149-155 has no h36m database for config TITLE %r' % cfg.TITLE)
    elif dataset == 'mpii3d':
        stem = 'mpii3d_val_scale12'
    else:
        raise ValueError('Wrong target dataset %r (3dpw | h36m | mpii3d)' % dataset)
    return osp.join(db_dir, stem + '_db.pt'), osp.join(db_dir, stem + '_pseudotheta.pt')


# What the reference publishes for this path (mm; accel in mm/s^2): asset/wpw.png (Table 1) and asset/wopw.png (Table 2) of the reference checkout,
# referenced at README.md:86-87; window T + 1 = 6 frames, 2-layer GRU, hidden 1024.  Keyed by (config TITLE, evaluate.py --dataset).
PUBLISHED_METRICS = {
    ('repr_wpw_3dpw_model', '3dpw'): {'mpjpe_pa': 52.3, 'mpjpe': 84.6, 'mpvpe': 100.3, 'accel_err': 11.4},
    ('repr_wpw_h36m_mpii3d_model', 'mpii3d'): {'mpjpe_pa': 63.1, 'mpjpe': 96.2, 'accel_err': 16.7},
    ('repr_wpw_h36m_mpii3d_model', 'h36m'): {'mpjpe_pa': 47.1, 'mpjpe': 68.6, 'accel_err': 12.1},
    ('repr_wopw_3dpw_model', '3dpw'): {'mpjpe_pa': 56.1, 'mpjpe': 93.9, 'mpvpe': 115.9, 'accel_err': 11.7},
    ('repr_wopw_mpii3d_model', 'mpii3d'): {'mpjpe_pa': 62.9, 'mpjpe': 99.5, 'accel_err': 17.2},
    ('repr_wopw_h36m_model', 'h36m'): {'mpjpe_pa': 41.2, 'mpjpe': 61.6, 'accel_err': 12.0},
}


def published_row(cfg_title, dataset):
    """The published accuracy row for an experiment config and evaluation set, or None."""
    return PUBLISHED_METRICS.get((cfg_title, dataset))


def compare_with_published(measured, cfg_title, dataset):
    """{metric: {'measured', 'published', 'diff'}} for the metrics both have (evaluate.py:461's keys); None without a published row."""
    row = published_row(cfg_title, dataset)
    if row is None or not measured:
        return None
    return {k: {'measured': float(measured[k]), 'published': v, 'diff': float(measured[k]) - v} for k, v in row.items() if k in measured}
