"""Generate tests/golden/*.npz by running the REFERENCE's own classes.

Run only in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

The reference (pure Python) cannot be imported as-is here: `yacs`, `torchvision`
and `smplx` are not installed and the licence-gated data files are absent
(SURVEY.md 8c).  This script injects stub modules for those three imports and
patches `np.load` for the two data files read at construction time, then runs the
real `lib.models.tepose.TePose`, `lib.models.spin.Regressor`,
`lib.utils.geometry.*` from /root/reference on deterministic synthetic weights
(tepose_amd.synth, regenerable anywhere) and stores the outputs.

The `smplx.SMPL` stub is the oracle's own LBS (oracle.tepose_ref.lbs): these
vectors therefore pin the encoder, FC loop, rot6d, the SMPL wrapper's joint
logic, the J_regressor path, projection and R->axis-angle against reference
code; LBS itself stays unpinned (see oracle/tepose_ref.py header).

Fixtures hold inputs' seeds and expected outputs only -- no reference source.
"""
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from tepose_amd import synth  # noqa: E402
from oracle import tepose_ref as O  # noqa: E402

SMPL_NP = synth.synthetic_smpl(0)
MEAN = synth.synthetic_mean_params(0)


class _StubSMPL(nn.Module):
    """Stand-in for smplx.SMPL: oracle LBS over the synthetic tables; returns the 45
    joints smplx would (24 posed + 21 vertex-picked)."""

    def __init__(self, *a, **k):
        super().__init__()
        self.t = O.smpl_tensors(SMPL_NP)

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=False, **kw):
        if pose2rot:      # compute_error_verts (lib/utils/eval_utils.py:155-169): axis-angle in, [N,3] + [N,69]
            R = O.batch_rodrigues(torch.cat([global_orient, body_pose], dim=1).reshape(-1, 3)).view(-1, 24, 3, 3)
            global_orient, body_pose = R[:, :1], R[:, 1:]
        else:
            R = torch.cat([global_orient, body_pose], dim=1)
        dt = R.dtype
        t = {k: (v.to(dt) if torch.is_tensor(v) else v) for k, v in self.t.items()}
        verts, posed = O.lbs(t, betas, R)
        joints = torch.cat([posed, verts[:, O.EXTRA_VERTEX_IDS]], dim=1)
        return _Out(vertices=verts, global_orient=global_orient, body_pose=body_pose,
                    joints=joints, betas=betas, full_pose=R)


_Out = namedtuple('SMPLOutput', 'vertices global_orient body_pose joints betas full_pose')


def install_stubs():
    sys.path.insert(0, REF)
    yc = types.ModuleType('yacs.config')

    class CN(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            import copy
            return copy.deepcopy(self)

    yc.CfgNode = CN
    y = types.ModuleType('yacs')
    y.config = yc
    sys.modules.update({'yacs': y, 'yacs.config': yc})
    tv, tvm, tvr = (types.ModuleType(n) for n in
                    ('torchvision', 'torchvision.models', 'torchvision.models.resnet'))
    tv.models = tvm
    tvm.resnet = tvr
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm, 'torchvision.models.resnet': tvr})
    sx, sxb, sxl = (types.ModuleType(n) for n in ('smplx', 'smplx.body_models', 'smplx.lbs'))
    sx.SMPL = _StubSMPL
    sxb.SMPLOutput = _Out
    sxl.vertices2joints = lambda J, v: torch.einsum('bik,ji->bjk', [v, J])
    sys.modules.update({'smplx': sx, 'smplx.body_models': sxb, 'smplx.lbs': sxl})
    _np_load = np.load

    def fake_load(p, *a, **k):
        s = str(p)
        if s.endswith('J_regressor_extra.npy'):
            return SMPL_NP['J_regressor_extra']
        if s.endswith('smpl_mean_params.npz'):
            # a fresh copy per load, as a real np.load gives: Regressor.__init__ wraps the arrays with torch.from_numpy, so two
            # models built from ONE dict would share their init_pose / init_shape / init_cam buffers (load_state_dict of the second
            # would overwrite the first's)
            return {k: np.array(v, copy=True) for k, v in MEAN.items()}
        return _np_load(p, *a, **k)

    np.load = fake_load


def verts_digest(v):
    """[B,6890,3] -> strided subsample + per-person checksums (keeps fixtures small)."""
    v = np.asarray(v, dtype=np.float64)
    return {'verts_sub': v[:, ::53].astype(np.float32),
            'verts_sum': v.sum(axis=1).astype(np.float64),
            'verts_l2': np.sqrt((v * v).sum(axis=(1, 2))).astype(np.float64)}


def run_case(T_mod, name, L, H, B, T, use_jreg, seed_w=0, seed_x=1234):
    model = T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval()
    sd_np = synth.synthetic_state_dict(L, H, seed_w)
    sd = model.state_dict()
    for k in sd:
        if k in sd_np:
            assert tuple(sd[k].shape) == sd_np[k].shape, (k, sd[k].shape, sd_np[k].shape)
            sd[k] = torch.from_numpy(sd_np[k])
    missing = [k for k in sd_np if k not in sd]
    assert not missing, missing
    model.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synth.synthetic_windows(B, T, seed_x))
    J = torch.from_numpy(SMPL_NP['J_regressor_h36m']) if use_jreg else None
    with torch.no_grad():
        feat = model.encoder(x)
        out = model(x, J_regressor=J)[0]
        feat_tr = model.encoder(x, is_train=True)
    d = {'meta': np.array([L, H, B, T, int(use_jreg), seed_w, seed_x], dtype=np.int64),
         'feature': feat.numpy(), 'feature_train': feat_tr.numpy(),
         'theta': out['theta'].numpy(), 'kp_2d': out['kp_2d'].numpy(),
         'kp_3d': out['kp_3d'].numpy(), 'rotmat': out['rotmat'].numpy()}
    d.update(verts_digest(out['verts'].numpy()))
    # regressor internals through the reference module on the same feature
    with torch.no_grad():
        reg = model.regressor
        pose, shape, cam = reg.init_pose.expand(B, -1), reg.init_shape.expand(B, -1), reg.init_cam.expand(B, -1)
        for _ in range(3):
            xc = torch.cat([feat, pose, shape, cam], 1)
            xc = reg.fc2(reg.fc1(xc))
            pose, shape, cam = reg.decpose(xc) + pose, reg.decshape(xc) + shape, reg.deccam(xc) + cam
    d['pose6d'] = pose.numpy()
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print('wrote', name, {k: v.shape for k, v in d.items()})


def regressor_init_case(P_mod, name, N, n_iter, use_jreg, seed_w=3):
    """The reference's Regressor.forward with per-call init_pose / init_shape / init_cam and a non-default n_iter
    (lib/models/spin.py:240-251): the API the reference offers beyond what TePose.forward uses."""
    reg = P_mod.Regressor().eval()
    sd_np = synth.synthetic_state_dict(1, 64, seed_w)
    sd = reg.state_dict()
    for k in sd:
        kk = 'regressor.' + k
        if kk in sd_np:
            assert tuple(sd[k].shape) == sd_np[kk].shape, (k, sd[k].shape, sd_np[kk].shape)
            sd[k] = torch.from_numpy(sd_np[kk])
    reg.load_state_dict(sd, strict=True)
    feat = torch.from_numpy(synth.normal('gold/feat%d' % N, (N, 2048), std=0.5))
    ip = torch.from_numpy(synth.normal('gold/ip%d' % N, (N, 144), std=0.7))
    ish = torch.from_numpy(synth.normal('gold/is%d' % N, (N, 10), std=0.5))
    ic = torch.from_numpy(synth.normal('gold/ic%d' % N, (N, 3), std=0.1)) + torch.tensor([0.9, 0., 0.])
    J = torch.from_numpy(SMPL_NP['J_regressor_h36m']) if use_jreg else None
    with torch.no_grad():
        out = reg(feat, init_pose=ip, init_shape=ish, init_cam=ic, n_iter=n_iter, J_regressor=J)[0]
        only_pose = reg(feat, init_pose=ip, n_iter=n_iter, J_regressor=J)[0]       # the other two from the mean parameters
    d = {'meta': np.array([N, n_iter, int(use_jreg), seed_w], dtype=np.int64),
         'theta': out['theta'].numpy(), 'kp_2d': out['kp_2d'].numpy(), 'kp_3d': out['kp_3d'].numpy(),
         'rotmat': out['rotmat'].numpy(), 'theta_only_pose': only_pose['theta'].numpy(),
         'kp_3d_only_pose': only_pose['kp_3d'].numpy()}
    d.update(verts_digest(out['verts'].numpy()))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print('wrote', name, {k: v.shape for k, v in d.items()})


def driver_case(T_mod, name, L, H, N, T, seed_w, seed_x):
    """One clip through the reference's own autoregressive loop (evaluate.py:247-269,
    written out here as the caller does it, model = the reference TePose)."""
    model = T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval()
    sd_np = synth.synthetic_state_dict(L, H, seed_w)
    sd = model.state_dict()
    for k in sd:
        if k in sd_np:
            sd[k] = torch.from_numpy(sd_np[k])
    model.load_state_dict(sd, strict=True)
    w = synth.synthetic_windows(1, N, seed_x)[0]           # [N,2133]: features + plausible thetas
    feat = torch.from_numpy(w[:, :2048].copy())
    theta_input = torch.from_numpy(w[:T - 1, 2048:].copy())
    theta_input[:, :3] = torch.tensor([1., 0., 0.])        # evaluate.py:177-178
    theta0 = theta_input.clone()
    J = torch.from_numpy(SMPL_NP['J_regressor_h36m'])
    th, kp, vs = [], [], []
    with torch.no_grad():
        for j in range(N - T + 1):
            inp = torch.zeros((1, T, 2048 + 85))
            inp[0, :, :2048] = feat[None, j:j + T, :].clone()
            inp[0, :T - 1, 2048:] = theta_input.clone()
            preds = model(inp, J_regressor=J, is_train=False)
            th.append(preds[-1]['theta'].view(-1, 85).numpy().copy())
            kp.append(preds[-1]['kp_3d'].view(-1, 14, 3).numpy().copy())
            vs.append(preds[-1]['verts'].view(-1, 6890, 3)[:, ::53].numpy().copy())
            theta_input[:T - 2, :] = theta_input[1:T - 1, :].clone()
            theta_input[T - 2, :] = preds[-1]['theta'].clone().detach()
    np.savez_compressed(os.path.join(HERE, name + '.npz'),
                        meta=np.array([L, H, N, T, seed_w, seed_x], dtype=np.int64), theta_init=theta0.numpy(),
                        theta=np.concatenate(th), kp_3d=np.concatenate(kp), verts_sub=np.concatenate(vs))
    print('wrote', name, np.concatenate(th).shape)


def padded_case(T_mod, name, L, H, lens, T, seed_w, seed_x):
    """The batched whole-clip validation loop of lib/core/trainer.py:307-357 on one padded batch as the validation
    Datasets emit it (lib/dataset/threedpw_test.py:62-134: clips zero-padded to the longest one, arrays staged in
    float16, theta_pseu with cam = [1, 0, 0], `vidlen_each`, `index`): every clip advances through ALL vidlen - T + 1
    windows, padding included, the accumulators keep rows with j < vidlen_each - T + 1.  Loop written out as the caller
    has it, model = the reference TePose."""
    model = T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval()
    sd_np = synth.synthetic_state_dict(L, H, seed_w)
    sd = model.state_dict()
    for k in sd:
        if k in sd_np:
            sd[k] = torch.from_numpy(sd_np[k])
    model.load_state_dict(sd, strict=True)
    C, vidlen = len(lens), max(lens)
    feats = np.zeros((C, vidlen, 2048), dtype=np.float16)
    theta_pseu = np.zeros((C, vidlen, 85), dtype=np.float16)
    for c, n in enumerate(lens):
        w = synth.synthetic_windows(1, n, seed_x + c)[0]
        feats[c, :n] = w[:, :2048]
        theta_pseu[c, :n] = np.concatenate([np.tile(np.array([1., 0., 0.], dtype=np.float32), (n, 1)), w[:, 2051:]], axis=1)
    target = {'features': torch.from_numpy(feats).float(), 'theta_pseu': torch.from_numpy(theta_pseu).float(),
              'vidlen_each': torch.tensor(lens).float().view(C, 1), 'index': torch.arange(C).float().view(C, 1)}
    kp3d = np.zeros((C, vidlen, 14, 3), dtype=np.float16)
    for c, n in enumerate(lens):
        kp3d[c, :n] = synth.normal('pad/kp%d_%d' % (seed_x, c), (n, 14, 3), std=0.3)
    target['kp_3d'] = torch.from_numpy(kp3d).float()
    J = torch.from_numpy(SMPL_NP['J_regressor_h36m'])
    acc_j3d, acc_theta, acc_verts, acc_tj3d = [], [], [], []
    with torch.no_grad():
        for j in range(vidlen - T + 1):
            if j == 0:
                theta_input = torch.zeros((C, vidlen, 85))
                theta_input[target['index'].view(-1).long(), :T - 1, :] = target['theta_pseu'][:, :T - 1, :]
                pred_j3d_tsr = torch.zeros((C, vidlen, 14, 3))
            inp = torch.zeros((C, T, 2048 + 85))
            inp[:, :, :2048] = target['features'][:, j:j + T, :]
            inp[:, :T - 1, 2048:] = theta_input[target['index'].view(-1).long(), j:j + T - 1, :]
            preds = model(inp, J_regressor=J)
            pred_j3d = preds[-1]['kp_3d'].view(-1, 14, 3)
            theta_input[target['index'].view(-1).long(), j + T - 1, :] = preds[-1]['theta']
            keep = j < (target['vidlen_each'].view(-1) - T + 1)
            acc_j3d.append(pred_j3d[keep].numpy().copy())
            acc_tj3d.append(target['kp_3d'][:, j + T - 1][keep].numpy().copy())
            acc_theta.append(preds[-1]['theta'].view(-1, 85)[keep].numpy().copy())
            acc_verts.append(preds[-1]['verts'].view(-1, 6890, 3)[keep][:, ::53].numpy().copy())
            pred_j3d_tsr[:, j + T - 1] = pred_j3d
    # Trainer.evaluate (trainer.py:437-488) on these accumulators, with the reference's own metric functions
    from lib.utils.eval_utils import batch_compute_similarity_transform_torch, compute_accel, compute_error_accel
    pj, tj = torch.from_numpy(np.concatenate(acc_j3d)).clone(), torch.from_numpy(np.concatenate(acc_tj3d)).clone()
    pj -= (pj[:, [2], :] + pj[:, [3], :]) / 2.0
    tj -= (tj[:, [2], :] + tj[:, [3], :]) / 2.0
    errors = torch.sqrt(((pj - tj) ** 2).sum(dim=-1)).mean(dim=-1).numpy()
    S1_hat = batch_compute_similarity_transform_torch(pj, tj)
    errors_pa = torch.sqrt(((S1_hat - tj) ** 2).sum(dim=-1)).mean(dim=-1).numpy()
    ptsr, ttsr = pred_j3d_tsr.clone(), target['kp_3d'].clone()
    ptsr -= (ptsr[:, :, [2], :] + ptsr[:, :, [3], :]) / 2.0
    # trainer.py:470 as written: `target_j3ds_tsr[:,[2],:]` on the [C, vidlen, J, 3] tensor indexes FRAMES 2 and 3, i.e. the
    # ground truth loses one constant per clip and joint, not its per-frame pelvis (its accelerations stay the raw ones)
    ttsr -= (ttsr[:, [2], :] + ttsr[:, [3], :]) / 2.0
    accel = compute_accel(ptsr, target['vidlen_each'], T) * 1000
    accel_err = compute_error_accel(joints_pred=ptsr, joints_gt=ttsr, vidlen_each=target['vidlen_each'], seqlen=T) * 1000
    evald = np.array([np.mean(errors) * 1000, np.mean(errors_pa) * 1000, float(accel), float(accel_err)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), kp_3d=kp3d, eval_mpjpe_pa_accel_accelerr=evald,
                        meta=np.array([L, H, T, seed_w, seed_x] + list(lens), dtype=np.int64),
                        features=feats, theta_pseu=theta_pseu, pred_j3d=np.concatenate(acc_j3d),
                        pred_theta=np.concatenate(acc_theta), pred_verts_sub=np.concatenate(acc_verts),
                        pred_j3d_tsr=pred_j3d_tsr.numpy())
    print('wrote', name, np.concatenate(acc_j3d).shape, pred_j3d_tsr.shape)


def vibe_case(name, L, H, B, N, seed_w, seed_x, bidirectional=False, add_linear=True, use_residual=True):
    """Reference lib.models.vibe.VIBE (GRU [+ relu + Linear] [+ residual], then the per-frame regressor).  The first two
    cases are the configuration evaluate.py:93-101 builds; the others cover the remaining constructor flags
    (vibe.py:27-65).  A model without a linear and hidden != 2048 has no regressor-compatible output: encoder only."""
    import lib.models.vibe as V_mod
    model = V_mod.VIBE(seqlen=N, n_layers=L, hidden_size=H, add_linear=add_linear, bidirectional=bidirectional,
                       use_residual=use_residual, pretrained='').eval()
    sd_np = synth.synthetic_vibe_state_dict(L, H, seed_w, bidirectional=bidirectional, add_linear=add_linear)
    sd = model.state_dict()
    assert {k for k in sd if k.startswith('encoder.')} == {k for k in sd_np if k.startswith('encoder.')}
    for k in sd_np:
        assert k in sd and tuple(sd[k].shape) == sd_np[k].shape, k
        sd[k] = torch.from_numpy(sd_np[k])
    model.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synth.synthetic_windows(B, N, seed_x)[:, :, :2048].copy())
    J = torch.from_numpy(SMPL_NP['J_regressor_h36m'])
    meta = np.array([L, H, B, N, seed_w, seed_x, int(bidirectional), int(add_linear), int(use_residual)], dtype=np.int64)
    with torch.no_grad():
        feat = model.encoder(x)
        if feat.shape[-1] != 2048:
            np.savez_compressed(os.path.join(HERE, name + '.npz'), meta=meta, feature=feat.numpy())
            print('wrote', name, feat.shape, '(encoder only)')
            return
        out = model(x, J_regressor=J)[-1]
    np.savez_compressed(os.path.join(HERE, name + '.npz'),
                        meta=meta[:6] if (not bidirectional and add_linear and use_residual) else meta, feature=feat.numpy(),
                        theta=out['theta'].numpy(), kp_3d=out['kp_3d'].numpy(), rotmat=out['rotmat'].numpy(),
                        verts_sub=out['verts'].numpy()[:, :, ::53])
    print('wrote', name, feat.shape, out['theta'].shape)


def metrics_case():
    """Reference metric code as evaluate.py:413-450 strings it together (eval_utils imports as-is)."""
    from lib.utils.eval_utils import batch_compute_similarity_transform_torch, compute_error_accel_eval
    from lib.data_utils._kp_utils import convert_kps
    out = {}
    for tag, J, mode in (('lsp14', 14, 0), ('mpii17', 17, 1)):
        target = torch.from_numpy(synth.normal('met/t' + tag, (60, J, 3), std=0.35))
        pred = target + torch.from_numpy(synth.normal('met/n' + tag, (60, J, 3), std=0.04))
        pred[:, :, 1] *= 1.07                                   # scale + rotation the alignment must remove
        pj, tj = pred.clone().float(), target.clone().float()
        if mode == 1:
            pp, tp = pj[:, [-3], :], tj[:, [-3], :]
        else:
            pp, tp = (pj[:, [2], :] + pj[:, [3], :]) / 2.0, (tj[:, [2], :] + tj[:, [3], :]) / 2.0
        pj -= pp
        tj -= tp
        mpjpe = torch.sqrt(((pj - tj) ** 2).sum(dim=-1)).numpy().mean(axis=-1) * 1000
        S1_hat = batch_compute_similarity_transform_torch(pj, tj)
        pa = torch.sqrt(((S1_hat - tj) ** 2).sum(dim=-1)).numpy().mean(axis=-1) * 1000
        accel = np.zeros(len(pj))
        accel[1:-1] = compute_error_accel_eval(joints_pred=pj.numpy(), joints_gt=tj.numpy()) * 1000
        out.update({tag + '_pred': pred.numpy(), tag + '_target': target.numpy(), tag + '_mpjpe': mpjpe,
                    tag + '_pa': pa, tag + '_accel': accel})
    code = np.arange(49, dtype=np.float64)[None, :, None].repeat(3, axis=2)
    out['spin_to_common'] = convert_kps(code, src='spin', dst='common')[0, :, 0].astype(np.int64)
    out['spin_to_mpii3d_test'] = convert_kps(code, src='spin', dst='mpii3d_test')[0, :, 0].astype(np.int64)
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **out)
    print('wrote metrics', out['spin_to_common'], out['spin_to_mpii3d_test'])


def eval_valid_i(name, lens):
    """Deterministic `valid_i` columns for a synthetic MPI-INF-3DHP database ([n, 1] per clip, as
    dataset_data['valid_i'][indexes][valids] in evaluate.py:199): leading, trailing and interior holes, one clip
    with no valid frame at all (evaluate.py:399-401 skips it) and one with a single valid frame (no accel entry,
    evaluate.py:441)."""
    out = []
    for c, n in enumerate(lens):
        v = (synth.uniform01('%s/valid_i%d' % (name, c), n) > 0.3).astype(np.float64)
        if c % 5 == 0:
            v[:2] = 0; v[-3:] = 0; v[n // 2] = 0         # holes at both ends and inside
        elif c % 5 == 1:
            v[0] = 1; v[-1] = 1                          # first and last frame valid: both dropped from accel
        elif c % 5 == 2:
            v[:] = 0                                     # no valid frame
        elif c % 5 == 3:
            v[:] = 0; v[n // 3] = 1                      # exactly one
        out.append(v.reshape(n, 1))
    return out


def eval_case(T_mod, name, dataset, L, H, T, lens, seed_w, seed_db, joints=49, invalid_frames=()):
    """The reference's evaluation flow end to end on a synthetic `*_db.pt` (evaluate.py:169-206 keyed clips, :214-269
    VIBE bootstrap + autoregressive windows, :394-457 joint conversion / valid_i filter / pelvis / MPJPE / PA-MPJPE /
    accel / MPVPE, :461 frame-weighted means), written out as the script has it with the reference's own TePose and VIBE
    classes, convert_kps, batch_compute_similarity_transform_torch, compute_error_accel_eval and compute_error_verts.
    `dataset` plays the role of the data path's name ('mpii3d' in data_path, target_dataset == '3dpw')."""
    import lib.models.vibe as V_mod
    from lib.data_utils._kp_utils import convert_kps
    from lib.utils.eval_utils import (batch_compute_similarity_transform_torch, compute_error_accel_eval,
                                      compute_error_verts)
    from tepose_amd.data import synthetic_eval_db
    model = T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval()
    sd_np = synth.synthetic_state_dict(L, H, seed_w)
    sd = model.state_dict()
    for k in sd:
        if k in sd_np:
            sd[k] = torch.from_numpy(sd_np[k])
    model.load_state_dict(sd, strict=True)
    model_vibe = V_mod.VIBE(seqlen=T, n_layers=L, hidden_size=H, add_linear=True, bidirectional=False,
                            use_residual=True, pretrained='').eval()
    vsd_np = synth.synthetic_vibe_state_dict(L, H, seed_w + 1)
    vsd = model_vibe.state_dict()
    for k in vsd_np:
        vsd[k] = torch.from_numpy(vsd_np[k])
    model_vibe.load_state_dict(vsd, strict=True)
    dataset_data, psetheta = synthetic_eval_db(list(lens), seed=seed_db, joints=joints)
    for f in invalid_frames:                                                 # the db's own `valid` column
        dataset_data['valid'][f] = 0
    if dataset == 'mpii3d':
        dataset_data['valid_i'] = np.concatenate(eval_valid_i(name, lens), axis=0)
    J_regressor = torch.from_numpy(SMPL_NP['J_regressor_h36m']).float()
    seqlen = T
    full_res = {}
    vid_name_list = dataset_data['vid_name']
    unique_names = np.unique(vid_name_list)
    data_keyed = {}
    psetheta = np.array(psetheta, copy=True)
    for idx in range(psetheta.shape[0]):
        psetheta[idx, :] = np.concatenate((np.array([1., 0., 0.]), psetheta[idx, 3:].copy()), axis=0)
    for u_n in unique_names:
        indexes = vid_name_list == u_n
        valids = dataset_data['valid'][indexes].astype(bool)
        data_keyed[u_n] = {'features': dataset_data['features'][indexes][valids],
                           'joints3D': dataset_data['joints3D'][indexes][valids],
                           'vid_name': dataset_data['vid_name'][indexes][valids],
                           'theta_pseu': psetheta[indexes][valids]}
        if dataset == 'mpii3d':
            data_keyed[u_n]['pose'] = np.zeros((len(valids), 72))
            data_keyed[u_n]['shape'] = np.zeros((len(valids), 10))
            data_keyed[u_n]['valid_i'] = dataset_data['valid_i'][indexes][valids]
            J_regressor = None
        else:
            data_keyed[u_n]['pose'] = dataset_data['pose'][indexes][valids]
            data_keyed[u_n]['shape'] = dataset_data['shape'][indexes][valids]
    per_clip = {}
    tot_num_pose = 0
    with torch.no_grad():
        for ci, seq_name in enumerate(data_keyed.keys()):
            curr_feat = torch.tensor(data_keyed[seq_name]['features'])
            theta_input = torch.from_numpy(data_keyed[seq_name]['theta_pseu'][:seqlen - 1, :]).float()
            vid_names = data_keyed[seq_name]['vid_name']
            if len(vid_names) < seqlen:
                continue
            pred_j3ds, pred_verts = [], []
            batch = curr_feat[:seqlen].clone().unsqueeze(0)
            output = model_vibe(batch, J_regressor=J_regressor)[-1]
            n_kp = output['kp_3d'].shape[-2]
            pred_j3ds.append(output['kp_3d'][0, :seqlen - 1].view(-1, n_kp, 3).numpy())
            pred_verts.append(output['verts'][0, :seqlen - 1].view(-1, 6890, 3).numpy())
            for curr_idx in range(len(vid_names) - seqlen + 1):
                input_feat = torch.zeros((1, seqlen, 2048 + 85)).float()
                input_feat[0, :, :2048] = curr_feat[None, curr_idx:curr_idx + seqlen, :].clone()
                input_feat[0, :seqlen - 1, 2048:] = theta_input.clone()
                preds = model(input_feat, J_regressor=J_regressor, is_train=False)
                n_kp = preds[-1]['kp_3d'].shape[-2]
                pred_j3ds.append(preds[-1]['kp_3d'].view(-1, n_kp, 3).numpy())
                pred_verts.append(preds[-1]['verts'].view(-1, 6890, 3).numpy())
                theta_input[:seqlen - 2, :] = theta_input[1:seqlen - 1, :].clone()
                theta_input[seqlen - 2, :] = preds[-1]['theta'].clone().detach()
            pred_j3ds = np.vstack(pred_j3ds)
            raw_pred = pred_j3ds.copy()
            target_j3ds = data_keyed[seq_name]['joints3D']
            pred_verts = torch.from_numpy(np.vstack(pred_verts))
            dummy_cam = np.repeat(np.array([[1., 0., 0.]]), len(target_j3ds), axis=0)
            target_theta = np.concatenate([dummy_cam, data_keyed[seq_name]['pose'], data_keyed[seq_name]['shape']],
                                          axis=1).astype(np.float32)
            target_j3ds, target_theta = target_j3ds[:len(pred_j3ds)], target_theta[:len(pred_j3ds)]
            if dataset == 'mpii3d':
                target_j3ds = convert_kps(target_j3ds, src='spin', dst='mpii3d_test')
                pred_j3ds = convert_kps(pred_j3ds, src='spin', dst='mpii3d_test')
                valid_map = data_keyed[seq_name]['valid_i'][:, 0].nonzero()[0]
                if valid_map.size == 0:
                    continue
                while True:
                    if valid_map[-1] >= len(pred_j3ds):
                        valid_map = valid_map[:-1]
                    else:
                        break
            elif target_j3ds.shape[1] == 49:
                target_j3ds = convert_kps(target_j3ds, src='spin', dst='common')
                valid_map = np.arange(len(target_j3ds))
            else:
                valid_map = np.arange(len(target_j3ds))
            pred_j3ds = torch.from_numpy(pred_j3ds).float()
            target_j3ds = torch.from_numpy(target_j3ds).float()
            tot_num_pose += len(valid_map)
            if dataset == 'mpii3d':
                pred_pelvis = pred_j3ds[:, [-3], :]
                target_pelvis = target_j3ds[:, [-3], :]
            else:
                pred_pelvis = (pred_j3ds[:, [2], :] + pred_j3ds[:, [3], :]) / 2.0
                target_pelvis = (target_j3ds[:, [2], :] + target_j3ds[:, [3], :]) / 2.0
            pred_j3ds -= pred_pelvis
            target_j3ds -= target_pelvis
            m2mm = 1000
            mpvpe = compute_error_verts(target_theta=torch.from_numpy(target_theta), pred_verts=pred_verts) * m2mm
            mpjpe_all = torch.sqrt(((pred_j3ds - target_j3ds) ** 2).sum(dim=-1)).cpu().numpy().mean(axis=-1) * m2mm
            mpjpe = mpjpe_all[valid_map]
            S1_hat = batch_compute_similarity_transform_torch(pred_j3ds, target_j3ds)
            pa_all = torch.sqrt(((S1_hat - target_j3ds) ** 2).sum(dim=-1)).cpu().numpy().mean(axis=-1) * m2mm
            mpjpe_pa = pa_all[valid_map]
            accel_all = np.zeros((len(pred_j3ds,)))
            accel_all[1:-1] = compute_error_accel_eval(joints_pred=pred_j3ds, joints_gt=target_j3ds) * m2mm
            pose_map = valid_map.copy()
            accel_map = np.zeros(0, dtype=np.int64)
            has_accel = 0
            if len(valid_map) > 1:
                if valid_map[0] == 0:
                    valid_map = valid_map[1:]
                if valid_map[-1] == len(accel_all) - 1:
                    valid_map = valid_map[:-1]
                accel_map = valid_map.copy()
                has_accel = 1
                full_res.setdefault('accel_err', []).append(accel_all[valid_map])
            full_res.setdefault('mpjpe', []).append(mpjpe)
            full_res.setdefault('mpjpe_pa', []).append(mpjpe_pa)
            if dataset == '3dpw':
                full_res.setdefault('mpvpe', []).append(mpvpe)
            per_clip[ci] = {'raw_pred': raw_pred.astype(np.float32), 'mpjpe_all': mpjpe_all, 'pa_all': pa_all,
                            'accel_all': accel_all, 'pose_map': pose_map.astype(np.int64),
                            'accel_map': accel_map.astype(np.int64), 'has_accel': has_accel, 'mpvpe': mpvpe}
    final = {k: float(np.mean(np.concatenate(v))) for k, v in full_res.items()}
    d = {'meta': np.array([L, H, T, seed_w, seed_db, joints] + list(lens), dtype=np.int64),
         'invalid_frames': np.array(list(invalid_frames), dtype=np.int64),
         'tot_num_pose': np.array(tot_num_pose), 'evaluated_clips': np.array(sorted(per_clip), dtype=np.int64),
         'final_keys': np.array(sorted(final)), 'final_values': np.array([final[k] for k in sorted(final)])}
    if dataset == 'mpii3d':
        d['valid_i'] = dataset_data['valid_i']
    for ci, r in per_clip.items():
        for k, v in r.items():
            d['clip%d_%s' % (ci, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **d)
    print('wrote', name, final, 'poses', tot_num_pose, 'clips', sorted(per_clip))


def filter_cases():
    """Reference OneEuroFilter driven as lib/utils/smooth_pose.py:28-58 drives it, and the reference's
    quaternion utilities strung together as evaluate.py:32-59 (smooth_pose_mat) does."""
    from lib.utils.one_euro_filter import OneEuroFilter
    import lib.utils.slerp_filter_utils as SF

    class _Numpy1(object):
        """The reference's bundled transformations.py calls numpy.array(x, copy=False), which meant
        "copy only if needed" in the NumPy 1.x it was written for and raises in NumPy 2: give that
        module (only) the old meaning."""

        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def array(obj, *a, **k):
            if k.get('copy', True) is False:
                k.pop('copy')
                return np.asarray(obj, *a, **k)
            return np.array(obj, *a, **k)

    SF.numpy = _Numpy1()
    quaternion_from_matrix, quaternion_matrix, quaternion_slerp = (SF.quaternion_from_matrix, SF.quaternion_matrix,
                                                                   SF.quaternion_slerp)
    pose = (synth.normal('flt/pose', (50, 24, 3), std=0.4) +
            0.3 * np.sin(np.arange(50, dtype=np.float32) / 6.0)[:, None, None]).astype(np.float32)
    f = OneEuroFilter(np.zeros_like(pose[0]), pose[0], min_cutoff=0.004, beta=0.7)
    hat = np.zeros_like(pose)
    hat[0] = pose[0]
    for idx, p in enumerate(pose[1:]):
        idx += 1
        hat[idx] = f(np.ones_like(p) * idx, p)
    R = O.batch_rodrigues(torch.from_numpy(pose.reshape(-1, 3))).view(50, 24, 3, 3).numpy().astype(np.float32)
    R[7] = R[7] + synth.normal('flt/noise', (24, 3, 3), std=1e-4)          # not exactly orthonormal
    allq = []
    for j in range(R.shape[1]):
        quats = np.array([quaternion_from_matrix(R[i, j, :, :]) for i in range(R.shape[0])])
        for q in range(1, quats.shape[0]):
            if np.linalg.norm(quats[q - 1] - quats[q], axis=0) > np.linalg.norm(quats[q - 1] + quats[q], axis=0):
                quats[q] = -quats[q]
        for q in range(1, quats.shape[0]):
            quats[q] = quaternion_slerp(quats[q - 1], quats[q], 0.3)
        allq.append(np.array([quaternion_matrix(i)[:3, :3] for i in quats]))
    np.savez_compressed(os.path.join(HERE, 'filters.npz'), pose=pose, pose_hat=hat, R=R,
                        R_smooth=np.stack(allq, axis=1))
    print('wrote filters', hat.shape, np.stack(allq, axis=1).shape)


def geometry_cases(G):
    """Edge vectors for R->aa (each quaternion branch, angle 0, angles near pi about
    each axis), rot6d->R (incl. degenerate input) and projection."""
    rs = []

    def rod(axis, ang):
        a = np.asarray(axis, dtype=np.float64)
        a = a / np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K

    rs.append(np.eye(3))
    for ax in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, -2, 3], [-1, 0.2, 0.1]):
        for ang in (1e-4, 0.3, 1.5, 2.2, 3.0, np.pi - 1e-3, np.pi, -2.5):
            rs.append(rod(ax, ang))
    rng = np.random.RandomState(7)
    for _ in range(200):
        rs.append(rod(rng.randn(3), rng.uniform(0, np.pi)))
    R = torch.from_numpy(np.stack(rs).astype(np.float32))
    aa = G.rotation_matrix_to_angle_axis(R.clone())
    x6 = torch.from_numpy(synth.normal('geom/6d', (64, 144)))
    x6[0, :6] = 0.0                       # degenerate: zero a1 and a2
    x6[1, :6] = torch.tensor([1., 2., 0., 0., 0., 0.])   # a2 parallel to a1
    R6 = G.rot6d_to_rotmat(x6.clone())
    np.savez_compressed(os.path.join(HERE, 'geometry.npz'), R=R.numpy(), aa=aa.numpy(),
                        x6=x6.numpy(), R6=R6.numpy())
    print('wrote geometry', R.shape, aa.shape, R6.shape)


def main():
    install_stubs()
    import lib.models.tepose as T_mod
    import lib.models.smpl as S_mod
    import lib.models.spin as P_mod
    import lib.utils.geometry as G
    # table parity: the oracle's literal tables vs the reference's
    assert [S_mod.JOINT_MAP[n] for n in S_mod.JOINT_NAMES] == O.JOINT_MAP_49
    assert S_mod.H36M_TO_J14 == O.H36M_TO_J14
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if 'eval' in sys.argv[1:] or len(sys.argv) == 1:
        # BASELINE config 4's three evaluation sets (evaluate.py:394-457 branches)
        eval_case(T_mod, 'eval_mpii3d_L1H64_T5', 'mpii3d', 1, 64, 5, [13, 9, 8, 11, 12, 4, 10], 21, 31)
        eval_case(T_mod, 'eval_h36m_L1H64_T5', 'h36m', 1, 64, 5, [9, 3, 12], 22, 32, invalid_frames=(2, 15, 23))
        eval_case(T_mod, 'eval_h36m14_L1H64_T4', 'h36m', 1, 64, 4, [7, 10], 23, 33, joints=14)
        eval_case(T_mod, 'eval_3dpw_L2H64_T6', '3dpw', 2, 64, 6, [10, 14, 6], 24, 34, invalid_frames=(0, 11))
    if len(sys.argv) > 1:
        return
    run_case(T_mod, 'tepose_L2H1024_B2T6_j14', 2, 1024, 2, 6, True)
    run_case(T_mod, 'tepose_L2H1024_B2T6_j49', 2, 1024, 2, 6, False)
    run_case(T_mod, 'tepose_L2H1024_B2T16_j14', 2, 1024, 2, 16, True)
    run_case(T_mod, 'tepose_L2H1024_B1T32_j14', 2, 1024, 1, 32, True)
    run_case(T_mod, 'tepose_L1H128_B3T5_j49', 1, 128, 3, 5, False, seed_w=3, seed_x=77)
    run_case(T_mod, 'tepose_L3H64_B2T4_j14', 3, 64, 2, 4, True, seed_w=4, seed_x=78)
    driver_case(T_mod, 'driver_L2H128_N40T6', 2, 128, 40, 6, 6, 555)
    driver_case(T_mod, 'driver_L1H64_N9T4', 1, 64, 9, 4, 7, 556)
    padded_case(T_mod, 'padded_L2H128_T5', 2, 128, [23, 9, 17, 5], 5, 14, 700)
    regressor_init_case(P_mod, 'regressor_init_N5_it2_j14', 5, 2, True)
    regressor_init_case(P_mod, 'regressor_init_N3_it0_j49', 3, 0, False)
    vibe_case('vibe_L2H128_B2N20', 2, 128, 2, 20, 8, 901)
    vibe_case('vibe_L1H64_B1N5', 1, 64, 1, 5, 9, 902)
    vibe_case('vibe_bi_L2H64_B2N7', 2, 64, 2, 7, 10, 903, bidirectional=True, add_linear=False)
    vibe_case('vibe_bi_L1H100_B3N4_nores', 1, 100, 3, 4, 11, 904, bidirectional=True, add_linear=True, use_residual=False)
    vibe_case('vibe_nolin_L1H2048_B1N4', 1, 2048, 1, 4, 12, 905, add_linear=False)
    vibe_case('vibe_nolin_L2H96_B2N6', 2, 96, 2, 6, 13, 906, add_linear=False)
    metrics_case()
    filter_cases()
    geometry_cases(G)
    # projection vector
    j = torch.from_numpy(synth.normal('geom/j', (4, 14, 3), std=0.5))
    cam = torch.from_numpy(synth.normal('geom/cam', (4, 3), std=0.1)) + torch.tensor([0.9, 0., 0.])
    np.savez_compressed(os.path.join(HERE, 'projection.npz'), joints=j.numpy(), cam=cam.numpy(),
                        kp_2d=P_mod.projection(j, cam).numpy())


if __name__ == '__main__':
    main()
