"""Deterministic synthetic parameters and inputs for the TePose hot path.

None of the licence-gated files the reference reads (SMPL_NEUTRAL.pkl,
smpl_mean_params.npz, J_regressor_extra.npy, J_regressor_h36m.npy, checkpoints;
reference README.md:21-25, lib/models/smpl.py:54-56, lib/models/spin.py:232-235)
exist in this repo, so tests, goldens and bench.py use parameters of the true
shapes produced by a counter-based generator (splitmix64 over the element
index).  The generator depends on nothing but integer arithmetic, so the GPU box
regenerates bit-identical weights without shipping any blob.

Shapes follow SURVEY.md Appendix B (state-dict keys) and A.4 (SMPL tables).
"""
import functools
import math
from collections import OrderedDict

import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
FEAT_DIM = 2048
THETA_DIM = 85
INPUT_DIM = FEAT_DIM + THETA_DIM  # 2133, reference lib/models/tepose.py:54
NPOSE = 24 * 6

# kinematic tree of SMPL
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int64)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser over a uint64 array (wraps mod 2**64)."""
    with np.errstate(over='ignore'):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(tag):
    """64-bit stream id from a string tag (FNV-1a, then one splitmix round)."""
    h = 0xCBF29CE484222325
    for c in tag.encode():
        h = ((h ^ c) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return _splitmix64(np.array([h], dtype=np.uint64))[0]


def uniform01(tag, n):
    """n float64 values in [0,1) with 24 random bits each (exact in float32)."""
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + _stream(tag)
    bits = _splitmix64(idx) >> np.uint64(40)
    return bits.astype(np.float64) * (1.0 / (1 << 24))


def uniform(tag, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(tag, n)).astype(np.float32).reshape(shape)


def normal(tag, shape, std=1.0, mean=0.0):
    """Box-Muller on two uniform streams."""
    n = int(np.prod(shape))
    u1 = uniform01(tag + '/u1', n)
    u2 = uniform01(tag + '/u2', n)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (mean + std * r * np.cos(2.0 * math.pi * u2)).astype(np.float32).reshape(shape)


def _sparse_rows(tag, rows, cols, nnz):
    """Row-stochastic [rows, cols] matrix with nnz positive entries per row."""
    m = np.zeros((rows, cols), dtype=np.float64)
    pos = (uniform01(tag + '/pos', rows * nnz) * cols).astype(np.int64).reshape(rows, nnz)
    val = uniform01(tag + '/val', rows * nnz).reshape(rows, nnz) + 0.05
    for r in range(rows):
        np.add.at(m[r], pos[r], val[r])
    m /= m.sum(axis=1, keepdims=True)
    return m.astype(np.float32)


def synthetic_smpl(seed=0, skin_nnz=4):
    """Synthetic SMPL tables with the real model's shapes and structure.

    v_template[6890,3], shapedirs[6890,3,10], posedirs[207,20670] (smplx layout,
    SURVEY.md A.4), J_regressor[24,6890] (rows sum to 1), lbs_weights[6890,24]
    (<=4 non-zeros per vertex, rows sum to 1, like the real model),
    parents[24], J_regressor_extra[9,6890] (lib/models/smpl.py:67),
    J_regressor_h36m[17,6890] (evaluate.py:109).
    """
    t = 'smpl%d/' % seed
    v = normal(t + 'v', (NUM_VERTS, 3)) * np.array([0.30, 0.55, 0.15], dtype=np.float32)
    shapedirs = normal(t + 'S', (NUM_VERTS, 3, 10), std=0.012)
    posedirs = normal(t + 'P', (207, NUM_VERTS * 3), std=0.006)
    jreg = _sparse_rows(t + 'J', NUM_JOINTS, NUM_VERTS, 40)
    # skin weights: up to 4 joints per vertex
    w = np.zeros((NUM_VERTS, NUM_JOINTS), dtype=np.float64)
    jidx = (uniform01(t + 'Wj', NUM_VERTS * skin_nnz) * NUM_JOINTS).astype(np.int64).reshape(NUM_VERTS, skin_nnz)
    jval = uniform01(t + 'Wv', NUM_VERTS * skin_nnz).reshape(NUM_VERTS, skin_nnz) ** 2 + 1e-3
    for k in range(skin_nnz):
        np.add.at(w, (np.arange(NUM_VERTS), jidx[:, k]), jval[:, k])
    w /= w.sum(axis=1, keepdims=True)
    return OrderedDict(
        v_template=v.astype(np.float32),
        shapedirs=shapedirs,
        posedirs=posedirs,
        J_regressor=jreg,
        lbs_weights=w.astype(np.float32),
        parents=SMPL_PARENTS.copy(),
        J_regressor_extra=_sparse_rows(t + 'Jx', 9, NUM_VERTS, 24),
        J_regressor_h36m=_sparse_rows(t + 'Jh', 17, NUM_VERTS, 32),
    )


def synthetic_mean_params(seed=0):
    """Stand-in for smpl_mean_params.npz (lib/models/spin.py:232-235):
    pose[144] f32 (6D, interleaved layout of geometry.py:330-344 -> identity is
    [1,0,0,1,0,0]), shape[10] f64, cam[3] f32."""
    t = 'mean%d/' % seed
    pose = np.tile(np.array([1, 0, 0, 1, 0, 0], dtype=np.float32), 24) + normal(t + 'pose', (NPOSE,), std=0.15)
    shape = normal(t + 'shape', (10,), std=0.3).astype(np.float64)
    cam = np.array([0.9, 0.0, 0.0], dtype=np.float32)
    return {'pose': pose.astype(np.float32), 'shape': shape, 'cam': cam}


def encoder_param_shapes(n_layers, hidden):
    """Ordered (key, shape) list of encoder.* state-dict entries
    (reference lib/models/tepose.py:53-69)."""
    H = hidden
    out = []
    for l in range(n_layers):
        k_in = INPUT_DIM if l == 0 else H
        out += [('encoder.gru_fwd.weight_ih_l%d' % l, (3 * H, k_in)),
                ('encoder.gru_fwd.weight_hh_l%d' % l, (3 * H, H)),
                ('encoder.gru_fwd.bias_ih_l%d' % l, (3 * H,)),
                ('encoder.gru_fwd.bias_hh_l%d' % l, (3 * H,))]
    for l in range(n_layers):
        k_in = INPUT_DIM if l == 0 else 2 * H
        for sfx in ('', '_reverse'):
            out += [('encoder.gru_rec.weight_ih_l%d%s' % (l, sfx), (3 * H, k_in)),
                    ('encoder.gru_rec.weight_hh_l%d%s' % (l, sfx), (3 * H, H)),
                    ('encoder.gru_rec.bias_ih_l%d%s' % (l, sfx), (3 * H,)),
                    ('encoder.gru_rec.bias_hh_l%d%s' % (l, sfx), (3 * H,))]
    out += [('encoder.linear_fwd.weight', (2048, H)), ('encoder.linear_fwd.bias', (2048,)),
            ('encoder.linear_rec.weight', (2048, 2 * H)), ('encoder.linear_rec.bias', (2048,))]
    return out


def regressor_param_shapes():
    """regressor.* entries (reference lib/models/spin.py:215-238)."""
    return [('regressor.fc1.weight', (1024, FEAT_DIM + NPOSE + 13)), ('regressor.fc1.bias', (1024,)),
            ('regressor.fc2.weight', (1024, 1024)), ('regressor.fc2.bias', (1024,)),
            ('regressor.decpose.weight', (NPOSE, 1024)), ('regressor.decpose.bias', (NPOSE,)),
            ('regressor.decshape.weight', (10, 1024)), ('regressor.decshape.bias', (10,)),
            ('regressor.deccam.weight', (3, 1024)), ('regressor.deccam.bias', (3,))]


def synthetic_state_dict(n_layers=2, hidden=1024, seed=0, dec_gain=0.35):
    """(A pure function of its arguments: the last few results are memoised -- the published architecture's 64 M values take seconds to
    hash -- and handed out as fresh copies, so callers may scale or overwrite what they get.)
    Weights at PyTorch-default scale: U(+-1/sqrt(H)) for GRU, U(+-1/sqrt(fan_in))
    for Linear.  The decoders use `dec_gain`/sqrt(fan_in) instead of the
    reference's xavier gain 0.01 (spin.py:222-224) so the three regressor
    iterations move the pose by a visible amount and exercise rot6d/LBS with
    generic rotations.  Returns OrderedDict[str, np.float32 array] with the
    reference's key names, including the regressor buffers."""
    return OrderedDict((k, v.copy()) for k, v in _state_dict_cached(int(n_layers), int(hidden), int(seed), float(dec_gain)).items())


@functools.lru_cache(maxsize=3)
def _state_dict_cached(n_layers, hidden, seed, dec_gain):
    sd = OrderedDict()
    H = hidden
    for key, shape in encoder_param_shapes(n_layers, H):
        if '.gru_' in key:
            bound = 1.0 / math.sqrt(H)
        else:
            fan_in = H if 'linear_fwd' in key else 2 * H
            bound = 1.0 / math.sqrt(fan_in)
        sd[key] = uniform('sd%d/%s' % (seed, key), shape, -bound, bound)
    for key, shape in regressor_param_shapes():
        fan_in = {'fc1': FEAT_DIM + NPOSE + 13}.get(key.split('.')[1], 1024)
        bound = 1.0 / math.sqrt(fan_in)
        if '.dec' in key:
            bound *= dec_gain
        sd[key] = uniform('sd%d/%s' % (seed, key), shape, -bound, bound)
    mp = synthetic_mean_params(seed)
    sd['regressor.init_pose'] = mp['pose'][None].astype(np.float32)
    sd['regressor.init_shape'] = mp['shape'][None].astype(np.float32)
    sd['regressor.init_cam'] = mp['cam'][None].astype(np.float32)
    return sd


def synthetic_windows(B, T, seed=1234):
    """[B,T,2133] fp32 windows per SURVEY.md 8(d): post-ReLU-like features
    |N(0,1)|*0.5; theta slots cam=[0.9,0,0]+N(0,.05), pose aa N(0,.2), betas
    N(0,.5); the last frame's 85 theta slots are zero (evaluate.py:248-252)."""
    t = 'win%d/' % seed
    x = np.zeros((B, T, INPUT_DIM), dtype=np.float32)
    x[:, :, :FEAT_DIM] = np.abs(normal(t + 'f', (B, T, FEAT_DIM))) * 0.5
    th = np.concatenate([
        normal(t + 'cam', (B, T, 3), std=0.05) + np.array([0.9, 0, 0], dtype=np.float32),
        normal(t + 'aa', (B, T, 72), std=0.2),
        normal(t + 'beta', (B, T, 10), std=0.5)], axis=-1)
    x[:, :T - 1, FEAT_DIM:] = th[:, :T - 1]
    return x


def synthetic_vibe_state_dict(n_layers=2, hidden=1024, seed=0, dec_gain=0.35, bidirectional=False, add_linear=True):
    """State dict of the reference VIBE model (lib/models/vibe.py:27-101): encoder.gru.*,
    encoder.linear.* (present when bidirectional or add_linear, vibe.py:43-47), regressor.* -- same scales as
    synthetic_state_dict."""
    sd = OrderedDict()
    H = hidden
    D = 2 if bidirectional else 1
    b = 1.0 / math.sqrt(H)
    for l in range(n_layers):
        k_in = FEAT_DIM if l == 0 else D * H
        for sfx in ('', '_reverse')[:D]:
            for name, shape in (('weight_ih', (3 * H, k_in)), ('weight_hh', (3 * H, H)), ('bias_ih', (3 * H,)),
                                ('bias_hh', (3 * H,))):
                key = 'encoder.gru.%s_l%d%s' % (name, l, sfx)
                sd[key] = uniform('vibe%d/%s' % (seed, key), shape, -b, b)
    if bidirectional or add_linear:
        bl = 1.0 / math.sqrt(D * H)
        sd['encoder.linear.weight'] = uniform('vibe%d/lw' % seed, (2048, D * H), -bl, bl)
        sd['encoder.linear.bias'] = uniform('vibe%d/lb' % seed, (2048,), -bl, bl)
    full = synthetic_state_dict(1, 64, seed, dec_gain)
    for k, v in full.items():
        if k.startswith('regressor.'):
            sd[k] = v
    return sd
