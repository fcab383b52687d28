"""CPU ORACLE for the TePose per-window inference hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import it; the product path
(tepose_amd/) never does and fails loudly when the HIP extension is missing.

It restates, in plain torch-CPU tensor ops (any float dtype; float64 gives the
high-precision truth), the algorithm of the reference's `TePose.forward` in
eval mode.  Citations are relative to /root/reference:

  encoder      lib/models/tepose.py:44-87   (torch.nn.GRU gate math, gates r,z,n)
  regressor    lib/models/spin.py:240-291
  rot6d -> R   lib/utils/geometry.py:330-344
  SMPL wrapper lib/models/smpl.py:61-84
  projection   lib/models/spin.py:307-351
  R -> aa      lib/utils/geometry.py:68-233

PARITY PINNING.  Everything except LBS is pinned against the reference's own
code run in the build container (tests/golden/make_golden.py imports
/root/reference with stub modules and writes tests/golden/*.npz; the `-m "not
gpu"` tests check this file against those vectors): the model / function vectors
call the reference's classes and functions, the flow vectors (evaluation incl.
--filter, clip driver, demo live path, trainer validation from the database file
on, metrics, filters) EXECUTE the reference's own statements -- AST slices of
evaluate.py / demo.py, the unbound Trainer.validate / .evaluate, the validation
Dataset classes.  Axis-angle -> R is pinned to lib/utils/geometry.py:22-65.  The LBS arithmetic itself
lives in the third-party package `smplx` (requirements.txt:7 pins 0.1.13; the
code imports `SMPLOutput`, so a later 0.1.2x was really used), which is neither
vendored in the reference nor installed here and whose licence-gated model files
are absent: for `lbs()` below **parity is unpinned** -- it restates the published
SMPL / smplx.lbs algorithm (blend shapes, joint regression, pose blend shapes,
rigid transform chain, linear blend skinning) and is checked by
invariants only (identity pose, rigid global rotation, affine consistency).
"""
import torch
import torch.nn.functional as F

# ---- SMPL joint tables (values as in reference lib/models/smpl.py:14-58) ------------
# 49 output joints = indices into [24 LBS joints | 21 vertex-picked | 9 regressed].
JOINT_MAP_49 = [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 25, 26, 27, 28, 29, 30, 31,
                32, 33, 34, 8, 5, 45, 46, 4, 7, 21, 19, 17, 16, 18, 20, 47, 48, 49, 50, 51, 52,
                53, 24, 26, 25, 28, 27]
H36M_TO_J14 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]
# smplx VertexJointSelector for SMPL (face, feet, finger-tip vertices)
EXTRA_VERTEX_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                    2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]


def _t(a, dtype):
    return a.to(dtype) if torch.is_tensor(a) else torch.as_tensor(a, dtype=dtype)


# ---- encoder --------------------------------------------------------------------------
def gru_cell(gi, h, w_hh, b_hh):
    """One GRU step given gi = x W_ih^T + b_ih.  torch.nn.GRU semantics, gate rows
    [r; z; n] (call sites lib/models/tepose.py:53-64)."""
    H = h.shape[1]
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def _scan(seq, sd, prefix, steps=None):
    """Run one GRU direction over seq[T,B,K] in the given order; returns [steps,B,H]."""
    w_ih, w_hh = sd[prefix.replace('@', 'weight_ih')], sd[prefix.replace('@', 'weight_hh')]
    b_ih, b_hh = sd[prefix.replace('@', 'bias_ih')], sd[prefix.replace('@', 'bias_hh')]
    T, B = seq.shape[:2]
    h = seq.new_zeros(B, w_hh.shape[1])
    out = []
    for t in range(T if steps is None else steps):
        h = gru_cell(seq[t] @ w_ih.t() + b_ih, h, w_hh, b_hh)
        out.append(h)
    return torch.stack(out)


def encoder_fwd(sd, x, n_layers, is_train=False):
    """TemporalEncoder.forward (lib/models/tepose.py:71-87), computing only the
    cell steps whose results are consumed.  sd: dict of tensors
    keyed like the reference state dict without the 'encoder.' prefix."""
    xs = x.transpose(0, 1)                              # [T,B,F]  (tepose.py:73)
    seq = xs
    for l in range(n_layers):
        seq = _scan(seq, sd, 'gru_fwd.@_l%d' % l)
    y_last = seq[-1]                                    # y[-1]    (tepose.py:79)
    seq = xs.flip(0)                                    # torch.flip(x,[1]) (tepose.py:75)
    for l in range(n_layers):
        top = l == n_layers - 1
        f = _scan(seq, sd, 'gru_rec.@_l%d' % l, steps=1 if top else None)
        b = _scan(seq.flip(0), sd, 'gru_rec.@_l%d_reverse' % l).flip(0)
        if top:
            y_rec0 = torch.cat([f[0], b[0]], dim=1)     # y_rec[0] (tepose.py:80)
        else:
            seq = torch.cat([f, b], dim=2)
    y_fwd = F.relu(y_last) @ sd['linear_fwd.weight'].t() + sd['linear_fwd.bias']
    y_rec = F.relu(y_rec0) @ sd['linear_rec.weight'].t() + sd['linear_rec.bias']
    if is_train:
        return torch.stack([y_fwd, y_rec], dim=1)
    return (y_fwd + y_rec) / 2


def encoder_fwd_nn_gru(sd, x, n_layers, hidden):
    """Same function through torch.nn.GRU modules exactly as the reference builds
    them (all T steps of every direction) -- the op sequence the reference runs
    on CPU; used as the timed cpu_baseline and as a cross-check of encoder_fwd."""
    gf = torch.nn.GRU(x.shape[2], hidden, num_layers=n_layers, bidirectional=False)
    gr = torch.nn.GRU(x.shape[2], hidden, num_layers=n_layers, bidirectional=True)
    gf.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith('gru_fwd.')})
    gr.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith('gru_rec.')})
    gf, gr = gf.to(x.dtype), gr.to(x.dtype)
    with torch.no_grad():
        y, _ = gf(x.permute(1, 0, 2))
        y_rec, _ = gr(torch.flip(x, dims=[1]).permute(1, 0, 2))
        y_fwd = F.linear(F.relu(y[-1]), sd['linear_fwd.weight'], sd['linear_fwd.bias'])
        y_r = F.linear(F.relu(y_rec[0]), sd['linear_rec.weight'], sd['linear_rec.bias'])
    return (y_fwd + y_r) / 2


# ---- regressor FC loop ------------------------------------------------------------------
def regressor_iterations(sd, feat, n_iter=3, init=(None, None, None)):
    """Regressor.forward lines spin.py:243-261 (dropout = identity in eval); `init` = the optional per-call
    init_pose / init_shape / init_cam of spin.py:240-248."""
    B = feat.shape[0]
    pose = sd['init_pose'].expand(B, -1) if init[0] is None else init[0]
    shape = sd['init_shape'].expand(B, -1) if init[1] is None else init[1]
    cam = sd['init_cam'].expand(B, -1) if init[2] is None else init[2]
    for _ in range(n_iter):
        xc = torch.cat([feat, pose, shape, cam], 1)
        xc = xc @ sd['fc1.weight'].t() + sd['fc1.bias']
        xc = xc @ sd['fc2.weight'].t() + sd['fc2.bias']
        pose = xc @ sd['decpose.weight'].t() + sd['decpose.bias'] + pose
        shape = xc @ sd['decshape.weight'].t() + sd['decshape.bias'] + shape
        cam = xc @ sd['deccam.weight'].t() + sd['deccam.bias'] + cam
    return pose, shape, cam


# ---- geometry ---------------------------------------------------------------------------
def rot6d_to_rotmat(x):
    """geometry.py:330-344.  Interleaved 6D layout: a1 = x[0::2], a2 = x[1::2]."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-6)
    u = a2 - (b1 * a2).sum(dim=1, keepdim=True) * b1
    b2 = u / u.norm(dim=1, keepdim=True).clamp_min(1e-6)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=-1)


def rotmat_to_angle_axis(R):
    """geometry.py:68-233 on [N,3,3]: 4-branch quaternion of M = R^T, then
    quaternion -> axis-angle with atan2; NaN -> 0."""
    M = R.transpose(1, 2)
    m00, m01, m02 = M[:, 0, 0], M[:, 0, 1], M[:, 0, 2]
    m10, m11, m12 = M[:, 1, 0], M[:, 1, 1], M[:, 1, 2]
    m20, m21, m22 = M[:, 2, 0], M[:, 2, 1], M[:, 2, 2]
    d2 = m22 < 1e-6
    d01 = m00 > m11
    d0n1 = m00 < -m11
    t0 = 1 + m00 - m11 - m22
    q0 = torch.stack([m12 - m21, t0, m01 + m10, m20 + m02], -1)
    t1 = 1 - m00 + m11 - m22
    q1 = torch.stack([m20 - m02, m01 + m10, t1, m12 + m21], -1)
    t2 = 1 - m00 - m11 + m22
    q2 = torch.stack([m01 - m10, m20 + m02, m12 + m21, t2], -1)
    t3 = 1 + m00 + m11 + m22
    q3 = torch.stack([t3, m12 - m21, m20 - m02, m01 - m10], -1)
    c0 = (d2 & d01).to(R.dtype)[:, None]
    c1 = (d2 & ~d01).to(R.dtype)[:, None]
    c2 = (~d2 & d0n1).to(R.dtype)[:, None]
    c3 = (~d2 & ~d0n1).to(R.dtype)[:, None]
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0[:, None] * c0 + t1[:, None] * c1 + t2[:, None] * c2 + t3[:, None] * c3)
    q = q * 0.5
    w, xyz = q[:, 0], q[:, 1:]
    s2 = (xyz * xyz).sum(-1)
    s = torch.sqrt(s2)
    two_theta = 2.0 * torch.where(w < 0, torch.atan2(-s, -w), torch.atan2(s, w))
    k = torch.where(s2 > 0, two_theta / s, torch.full_like(s, 2.0))
    aa = xyz * k[:, None]
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def batch_rodrigues(aa):
    """smplx.lbs.batch_rodrigues [published algorithm]; pinned to the Rodrigues the reference itself holds,
    lib/utils/geometry.py:22-65 (quaternion form of the same map), by tests/golden/geometry.npz `rod_*` (2e-8 in fp64):
    angle = ||aa + 1e-8||, R = I + sin K + (1-cos) K^2."""
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    d = aa / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = torch.zeros_like(rx)
    K = torch.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=aa.dtype)[None]
    return eye + s * K + (1 - c) * (K @ K)


def projection(joints, cam):
    """spin.py:307-351 with R = I, centre = 0, f = 5000, then / (224/2)."""
    t = torch.stack([cam[:, 1], cam[:, 2], 2 * 5000. / (224. * cam[:, 0] + 1e-9)], dim=-1)
    p = joints + t[:, None]
    p = p / p[:, :, -1:]
    return (5000. * p[:, :, :2]) / (224. / 2.)


# ---- SMPL -------------------------------------------------------------------------------
def lbs(smpl, betas, rot_mats):
    """smplx.lbs.lbs with pose2rot=False [published algorithm; PARITY UNPINNED, see
    module docstring].  smpl: dict of tensors (v_template[V,3], shapedirs[V,3,10],
    posedirs[207,3V], J_regressor[24,V], lbs_weights[V,24], parents[24]).
    Returns verts[B,V,3], posed joints[B,24,3]."""
    B = betas.shape[0]
    dt = betas.dtype
    v_shaped = smpl['v_template'] + torch.einsum('bl,mkl->bmk', betas, smpl['shapedirs'])
    J = torch.einsum('bik,ji->bjk', v_shaped, smpl['J_regressor'])
    eye = torch.eye(3, dtype=dt)
    pose_feature = (rot_mats[:, 1:] - eye).reshape(B, -1)
    v_posed = v_shaped + (pose_feature @ smpl['posedirs']).view(B, -1, 3)
    parents = [int(p) for p in smpl['parents']]
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    Tm = torch.zeros(B, 24, 4, 4, dtype=dt)
    Tm[:, :, :3, :3] = rot_mats
    Tm[:, :, :3, 3] = rel
    Tm[:, :, 3, 3] = 1
    chain = [Tm[:, 0]]
    for i in range(1, 24):
        chain.append(chain[parents[i]] @ Tm[:, i])
    G = torch.stack(chain, dim=1)
    posed = G[:, :, :3, 3]
    A = G.clone()
    A[:, :, :3, 3] = G[:, :, :3, 3] - torch.einsum('bjik,bjk->bji', G[:, :, :3, :3], J)
    Tv = torch.einsum('vj,bjik->bvik', smpl['lbs_weights'], A)
    verts = torch.einsum('bvik,bvk->bvi', Tv[:, :, :3, :3], v_posed) + Tv[:, :, :3, 3]
    return verts, posed


def smpl_joints49(smpl, verts, posed):
    """smplx vertex joint selector (+21) then wrapper lib/models/smpl.py:72-84."""
    j45 = torch.cat([posed, verts[:, EXTRA_VERTEX_IDS]], dim=1)
    extra = torch.einsum('bik,ji->bjk', verts, smpl['J_regressor_extra'])
    return torch.cat([j45, extra], dim=1)[:, JOINT_MAP_49]


def regressor_fwd(sd, smpl, feat, J_regressor=None, n_iter=3, init=(None, None, None)):
    """Regressor.forward (spin.py:240-291) -> dict like one element of its list."""
    B = feat.shape[0]
    pose6d, shape, cam = regressor_iterations(sd, feat, n_iter, init)
    R = rot6d_to_rotmat(pose6d).view(B, 24, 3, 3)
    verts, posed = lbs(smpl, shape, R)
    joints = smpl_joints49(smpl, verts, posed)
    if J_regressor is not None:
        joints = torch.einsum('bik,ji->bjk', verts, J_regressor)[:, H36M_TO_J14]
    kp2d = projection(joints, cam)
    aa = rotmat_to_angle_axis(R.reshape(-1, 3, 3)).reshape(-1, 72)
    return {'theta': torch.cat([cam, aa, shape], dim=1), 'verts': verts, 'kp_2d': kp2d,
            'kp_3d': joints, 'rotmat': R, 'pose6d': pose6d}


def split_state_dict(state, dtype=torch.float32):
    """Full reference-keyed state dict -> (encoder dict, regressor dict) of tensors."""
    enc = {k[len('encoder.'):]: _t(v, dtype) for k, v in state.items() if k.startswith('encoder.')}
    reg = {k[len('regressor.'):]: _t(v, dtype) for k, v in state.items()
           if k.startswith('regressor.') and not k.startswith('regressor.smpl.')}
    return enc, reg


def smpl_tensors(smpl_np, dtype=torch.float32):
    out = {k: _t(v, dtype) for k, v in smpl_np.items() if k != 'parents'}
    out['parents'] = [int(p) for p in smpl_np['parents']]
    return out


def tepose_fwd(state, smpl_np, x, n_layers, J_regressor=None, dtype=torch.float32, nn_gru=False):
    """TePose.forward, eval mode (lib/models/tepose.py:121-136).  Returns the dict of
    the single list element plus 'feature' (encoder output) for module-level tests."""
    enc, reg = split_state_dict(state, dtype)
    smpl = smpl_tensors(smpl_np, dtype)
    x = _t(x, dtype)
    with torch.no_grad():
        if nn_gru:
            feat = encoder_fwd_nn_gru(enc, x, n_layers, enc['gru_fwd.weight_hh_l0'].shape[1])
        else:
            feat = encoder_fwd(enc, x, n_layers)
        out = regressor_fwd(reg, smpl, feat,
                            None if J_regressor is None else _t(J_regressor, dtype))
    out['feature'] = feat
    return out


def run_clip(state, smpl_np, feats, theta_init, seqlen, n_layers, J_regressor=None, dtype=torch.float32):
    """The per-clip autoregressive loop of evaluate.py:247-269 (B = 1, strictly serial):
    window j = frames j..j+T-1 with theta slots of its first T-1 frames = theta_input, last
    frame zero; afterwards theta_input shifts by one and takes the new prediction.
    Returns dict of stacked per-window outputs [N-T+1, ...]."""
    T = seqlen
    feats = _t(feats, dtype)
    theta_input = _t(theta_init, dtype).clone()
    enc, reg = split_state_dict(state, dtype)
    smpl = smpl_tensors(smpl_np, dtype)
    J = None if J_regressor is None else _t(J_regressor, dtype)
    outs = {'theta': [], 'kp_3d': [], 'verts': [], 'rotmat': []}
    with torch.no_grad():
        for j in range(feats.shape[0] - T + 1):
            x = torch.zeros(1, T, 2133, dtype=dtype)
            x[0, :, :2048] = feats[j:j + T]
            x[0, :T - 1, 2048:] = theta_input
            o = regressor_fwd(reg, smpl, encoder_fwd(enc, x, n_layers), J)
            for k in outs:
                outs[k].append(o[k][0])
            theta_input[:T - 2] = theta_input[1:T - 1].clone()
            theta_input[T - 2] = o['theta'][0]
    return {k: torch.stack(v) for k, v in outs.items()}


def vibe_encoder_fwd(sd, x, n_layers, use_residual=True):
    """VIBE TemporalEncoder.forward (lib/models/vibe.py:52-65): y = gru(x); when the module has a linear (bidirectional or
    add_linear, vibe.py:43-47) y = linear(relu(y)); y += x when use_residual and y is 2048 wide.  The configuration is read
    off the keys: gru.*_reverse => bidirectional, linear.weight => linear.  x [B,N,2048] -> [B,N,2048 or hidden]."""
    seq = x.transpose(0, 1)
    bidir = 'gru.weight_ih_l0_reverse' in sd
    for l in range(n_layers):
        f = _scan(seq, sd, 'gru.@_l%d' % l)
        seq = torch.cat([f, _scan(seq.flip(0), sd, 'gru.@_l%d_reverse' % l).flip(0)], dim=2) if bidir else f
    y = seq
    if 'linear.weight' in sd:
        y = F.relu(y) @ sd['linear.weight'].t() + sd['linear.bias']
    if use_residual and y.shape[-1] == 2048:
        y = y + x.transpose(0, 1)
    return y.transpose(0, 1)


def vibe_fwd(state, smpl_np, x, n_layers, J_regressor=None, dtype=torch.float32, use_residual=True):
    """VIBE.forward (lib/models/vibe.py:104-117): per-frame regressor over the encoder output."""
    enc, reg = split_state_dict(state, dtype)
    smpl = smpl_tensors(smpl_np, dtype)
    x = _t(x, dtype)
    with torch.no_grad():
        feat = vibe_encoder_fwd(enc, x, n_layers, use_residual)
        feat = feat.reshape(-1, feat.shape[-1])
        out = regressor_fwd(reg, smpl, feat, None if J_regressor is None else _t(J_regressor, dtype))
    out['feature'] = feat
    return out


# ---- metrics (evaluate.py:413-457; lib/utils/eval_utils.py) ---------------------------------
def procrustes_align(S1, S2):
    """batch_compute_similarity_transform_torch (eval_utils.py:287-337) on [N,J,3] inputs."""
    S1t, S2t = S1.permute(0, 2, 1), S2.permute(0, 2, 1)
    mu1, mu2 = S1t.mean(dim=-1, keepdim=True), S2t.mean(dim=-1, keepdim=True)
    X1, X2 = S1t - mu1, S2t - mu2
    var1 = (X1 ** 2).sum(dim=1).sum(dim=1)
    K = X1.bmm(X2.permute(0, 2, 1))
    U, s, Vh = torch.linalg.svd(K)
    V = Vh.transpose(1, 2)
    Z = torch.eye(3, dtype=S1.dtype).repeat(U.shape[0], 1, 1)
    Z[:, -1, -1] *= torch.sign(torch.det(U.bmm(V.permute(0, 2, 1))))
    R = V.bmm(Z.bmm(U.permute(0, 2, 1)))
    scale = torch.stack([torch.trace(x) for x in R.bmm(K)]) / var1
    t = mu2 - scale[:, None, None] * R.bmm(mu1)
    return (scale[:, None, None] * R.bmm(S1t) + t).permute(0, 2, 1)


def joint_metrics(pred, target, pelvis='lsp'):
    """Per-frame mpjpe, pa_mpjpe, accel (mm) as evaluate.py:419-450 computes them."""
    pred, target = pred.clone(), target.clone()
    if pelvis == 'lsp':
        pp, tp = (pred[:, [2]] + pred[:, [3]]) / 2.0, (target[:, [2]] + target[:, [3]]) / 2.0
    else:
        pp, tp = pred[:, [-3]], target[:, [-3]]
    pred, target = pred - pp, target - tp
    mpjpe = torch.sqrt(((pred - target) ** 2).sum(-1)).mean(-1) * 1000
    pa = torch.sqrt(((procrustes_align(pred, target) - target) ** 2).sum(-1)).mean(-1) * 1000
    accel = torch.zeros(pred.shape[0], dtype=pred.dtype)
    if pred.shape[0] > 2:
        ag = target[:-2] - 2 * target[1:-1] + target[2:]
        ap = pred[:-2] - 2 * pred[1:-1] + pred[2:]
        accel[1:-1] = torch.linalg.norm(ap - ag, dim=2).mean(1) * 1000
    return {'mpjpe': mpjpe, 'pa_mpjpe': pa, 'accel': accel}


def verts_from_theta(smpl_np, theta, dtype=torch.float32):
    """GT mesh of compute_error_verts (eval_utils.py:155-169): pose2rot=True [LBS unpinned]."""
    smpl = smpl_tensors(smpl_np, dtype)
    theta = _t(theta, dtype)
    R = batch_rodrigues(theta[:, 3:75].reshape(-1, 3)).view(-1, 24, 3, 3)
    return lbs(smpl, theta[:, 75:], R)[0]


# ---- temporal post-filters (lib/utils/one_euro_filter.py; evaluate.py:32-59) ---------------------
def one_euro_filter(x, min_cutoff=0.004, beta=0.7, d_cutoff=1.0):
    """OneEuroFilter driven as lib/utils/smooth_pose.py:28-58 does: t = frame index, x0 = x[0]."""
    import math
    import numpy as np
    x = np.asarray(x, dtype=np.float32)
    out = x.copy()
    x_prev, dx_prev = x[0], np.float32(0.0)
    t_e = np.float32(1.0)
    for i in range(1, len(x)):
        r = np.float32(2 * math.pi * d_cutoff) * t_e
        a_d = r / (r + 1)
        dx = (x[i] - x_prev) / t_e
        dx_hat = a_d * dx + (1 - a_d) * dx_prev
        cutoff = np.float32(min_cutoff) + np.float32(beta) * np.abs(dx_hat)
        r = np.float32(2 * math.pi) * cutoff * t_e
        a = r / (r + 1)
        x_hat = a * x[i] + (1 - a) * x_prev
        out[i] = x_hat
        x_prev, dx_prev = x_hat, dx_hat
    return out


def slerp_smooth(R, ratio=0.3):
    """smooth_pose_mat (evaluate.py:48-59): per joint, sign-continuous quaternions, then
    q_t = slerp(q_{t-1}, q_t, ratio), back to matrices.  R [N,J,3,3] (numpy, any float)."""
    import numpy as np
    R = np.asarray(R, dtype=np.float64)
    N, J = R.shape[:2]
    out = np.zeros_like(R)
    for j in range(J):
        q = np.zeros((N, 4))
        for t in range(N):
            m = R[t, j]
            tr = np.trace(m)
            if tr > 0:
                s = np.sqrt(tr + 1.0) * 2
                q[t] = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
            elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
                s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
                q[t] = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
            elif m[1, 1] > m[2, 2]:
                s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
                q[t] = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
            else:
                s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
                q[t] = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
            # best-fit quaternion = dominant eigenvector of K(R) (transformations.quaternion_from_matrix,
            # isprecise=False): two power steps on K/3 + I/3 from the direct estimate
            K = np.array([[m[0, 0] - m[1, 1] - m[2, 2], m[0, 1] + m[1, 0], m[0, 2] + m[2, 0], m[2, 1] - m[1, 2]],
                          [m[0, 1] + m[1, 0], m[1, 1] - m[0, 0] - m[2, 2], m[1, 2] + m[2, 1], m[0, 2] - m[2, 0]],
                          [m[0, 2] + m[2, 0], m[1, 2] + m[2, 1], m[2, 2] - m[0, 0] - m[1, 1], m[1, 0] - m[0, 1]],
                          [m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1], m[0, 0] + m[1, 1] + m[2, 2]]])
            A = K / 3.0 + np.eye(4) / 3.0
            v = q[t][[1, 2, 3, 0]]
            for _ in range(2):
                v = A @ v
                v /= np.linalg.norm(v)
            q[t] = v[[3, 0, 1, 2]]
        for t in range(1, N):                                   # quat_correct
            if np.linalg.norm(q[t - 1] - q[t]) > np.linalg.norm(q[t - 1] + q[t]):
                q[t] = -q[t]
        for t in range(1, N):                                   # quat_smooth
            q0, q1 = q[t - 1] / np.linalg.norm(q[t - 1]), q[t] / np.linalg.norm(q[t])
            d = float(np.dot(q0, q1))
            if abs(abs(d) - 1.0) < 8.881784197001252e-16:
                q[t] = q0
                continue
            if d < 0:
                d, q1 = -d, -q1
            ang = np.arccos(d)
            if abs(ang) < 8.881784197001252e-16:
                q[t] = q0
                continue
            q[t] = (q0 * np.sin((1 - ratio) * ang) + q1 * np.sin(ratio * ang)) / np.sin(ang)
        for t in range(N):
            w, x, y, z = q[t] * np.sqrt(2.0 / np.dot(q[t], q[t]))
            out[t, j] = [[1 - y * y - z * z, x * y - z * w, x * z + y * w],
                         [x * y + z * w, 1 - x * x - z * z, y * z - x * w],
                         [x * z - y * w, y * z + x * w, 1 - x * x - y * y]]
    return out
