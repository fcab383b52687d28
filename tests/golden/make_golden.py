"""Generate tests/golden/*.npz by running the REFERENCE's own code.

Run only in the build container (needs /root/reference; never on the GPU box -- the file is listed in
`.gpurunignore` and holds no reference text):

    python tests/golden/make_golden.py [--out DIR] [eval | flows]

Two kinds of fixture:

* model / function vectors: the reference's `lib.models.tepose.TePose`, `lib.models.spin.Regressor`,
  `lib.models.vibe.VIBE`, `lib.utils.geometry.*`, `lib.utils.eval_utils.*`, `lib.utils.smooth_pose.smooth_pose`
  imported from /root/reference and called on deterministic synthetic weights (tepose_amd.synth);
* FLOW vectors (evaluation, autoregressive driver, demo live path, trainer validation loop, metrics, slerp filter): the reference's
  own STATEMENTS.  `evaluate.py` is parsed with `ast`; its top level (imports, helper defs) is executed as a module
  namespace and the body of its `if __name__ == "__main__":` block is compiled and executed in line-range slices
  (`RefScript.run`) in that namespace, after this generator has put the script's external inputs there (`parse_args`,
  `joblib.load`, `tqdm`, a checkpoint file).  Values are read back through probe calls the AST transformer inserts after
  given source lines.  The trainer loop is the unbound `lib.core.trainer.Trainer.validate` / `.evaluate` called on a
  namespace object.  Nothing of the loop / filter / trim logic is written out here.

The reference (pure Python) cannot be imported as-is in this image: `yacs`, `torchvision`, `smplx`, `cv2`, `progress`,
`pyrender` ... are not installed, the licence-gated data files are absent and there is no GPU (the
scripts call `.cuda()`).  `install_stubs` injects stand-in modules for the missing imports, patches `np.load` for the
three data files read at construction time and makes `.cuda()` the identity.

The `smplx.SMPL` stub is the oracle's own LBS (oracle.tepose_ref.lbs): these vectors therefore pin the encoder, FC
loop, rot6d, the SMPL wrapper's joint logic, the J_regressor path, projection, R->axis-angle, axis-angle->R
(geometry.py:22-65) and every flow against reference code; LBS itself stays unpinned (oracle/tepose_ref.py header).

Fixtures hold inputs' seeds and expected outputs only -- no reference source.
"""
import ast
import copy
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = HERE

from tepose_amd import synth  # noqa: E402
from oracle import tepose_ref as O  # noqa: E402

SMPL_NP = synth.synthetic_smpl(0)
MEAN = synth.synthetic_mean_params(0)


def save(name, **d):
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)


class _StubSMPL(nn.Module):
    """Stand-in for smplx.SMPL: oracle LBS over the synthetic tables; returns the 45 joints smplx would (24 posed + 21
    vertex-picked).  `pose2rot` defaults to True as in smplx (lib/utils/smooth_pose.py relies on the default)."""

    def __init__(self, *a, **k):
        super().__init__()
        self.t = O.smpl_tensors(SMPL_NP)

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True, **kw):
        if pose2rot:      # axis-angle in: [N,3] + [N,69] (eval_utils.py:155-169) or [N,1,3] + [N,23,3] (smooth_pose.py:45-49)
            n = betas.shape[0]
            R = O.batch_rodrigues(torch.cat([global_orient.reshape(n, -1), body_pose.reshape(n, -1)], dim=1).reshape(-1, 3)).view(-1, 24, 3, 3)
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


class CN(dict):
    """yacs.config.CfgNode as lib/core/config.py and the scripts use it."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)


class _Meta(type):
    def __getattr__(cls, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return _Meta(k, (), {'__init__': lambda self, *a, **kw: None})


class _Anything(types.ModuleType):
    """A module of a package this image lacks and the flows never reach (cv2, pyrender ...): every attribute exists."""
    __path__ = []

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        v = _Meta(k, (), {'__init__': lambda self, *a, **kw: None})
        setattr(self, k, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    MISSING = ('torchvision', 'cv2', 'skimage', 'pytube', 'trimesh', 'pyrender', 'multi_person_tracker', 'matplotlib', 'chumpy', 'h5py',
               'tensorboardX', 'torchgeometry', 'yolov3', 'gdown')

    def find_spec(self, name, path=None, target=None):
        if name.split('.')[0] in self.MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Anything(spec.name)

    def exec_module(self, module):
        pass


class _Bar(object):
    """progress.bar.Bar as lib/core/trainer.py:308,352-358 touches it."""
    elapsed_td = eta_td = suffix = ''

    def __init__(self, *a, **k):
        pass

    def next(self):
        pass

    def finish(self):
        pass


def install_stubs():
    sys.path.insert(0, REF)
    yc = types.ModuleType('yacs.config')
    yc.CfgNode = CN
    y = types.ModuleType('yacs')
    y.config = yc
    sys.modules.update({'yacs': y, 'yacs.config': yc})
    sx, sxb, sxl = (types.ModuleType(n) for n in ('smplx', 'smplx.body_models', 'smplx.lbs'))
    sx.SMPL = _StubSMPL
    sxb.SMPLOutput = _Out
    sxl.vertices2joints = lambda J, v: torch.einsum('bik,ji->bjk', [v, J])
    sys.modules.update({'smplx': sx, 'smplx.body_models': sxb, 'smplx.lbs': sxl})
    pg, pgb = types.ModuleType('progress'), types.ModuleType('progress.bar')
    pg.bar = pgb
    pgb.Bar = _Bar
    sys.modules.update({'progress': pg, 'progress.bar': pgb})
    sys.meta_path.append(_Finder())                     # last: real packages win
    _np_load = np.load

    def fake_load(p, *a, **k):
        s = str(p)
        if s.endswith('J_regressor_extra.npy'):
            return SMPL_NP['J_regressor_extra']
        if s.endswith('J_regressor_h36m.npy'):
            return np.array(SMPL_NP['J_regressor_h36m'], copy=True)
        if s.endswith('smpl_mean_params.npz'):
            # a fresh copy per load, as a real np.load gives: Regressor.__init__ wraps the arrays with torch.from_numpy, so two
            # models built from ONE dict would share their init_pose / init_shape / init_cam buffers (load_state_dict of the second
            # would overwrite the first's)
            return {k: np.array(v, copy=True) for k, v in MEAN.items()}
        return _np_load(p, *a, **k)

    np.load = fake_load
    # no GPU in the build container: the scripts' `.cuda()` calls keep the module where it is and COPY the tensor, as a host-to-device transfer does
    # (an identity would alias `theta_input` with the database rows it was sliced from, evaluate.py:219, and the loop's in-place shifts would edit them)
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()
    nn.Module.cuda = lambda self, *a, **k: self


# ---- the reference's scripts as executable AST ------------------------------------------------------------------------
def _is_main_guard(node):
    return (isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Name)
            and node.test.left.id == '__name__')


def _slice(stmts, lo, hi):
    """The statements with lo <= lineno and end_lineno <= hi, at the nesting level where they live."""
    here = [s for s in stmts if s.lineno >= lo and s.end_lineno <= hi]
    if here:
        return here
    for s in stmts:
        if s.lineno <= lo and s.end_lineno >= hi:
            for field in ('body', 'orelse', 'finalbody'):
                sub = getattr(s, field, None)
                if sub:
                    got = _slice(sub, lo, hi)
                    if got:
                        return got
    return []


class _Probes(ast.NodeTransformer):
    """Insert `__probe__(line)` after every statement that ENDS on one of the given lines."""

    def __init__(self, lines):
        self.lines = set(lines)
        self.placed = set()

    def generic_visit(self, node):
        super().generic_visit(node)
        for field in ('body', 'orelse', 'finalbody'):
            stmts = getattr(node, field, None)
            if not isinstance(stmts, list) or not stmts or not isinstance(stmts[0], ast.stmt):
                continue
            out = []
            for s in stmts:
                out.append(s)
                if s.end_lineno in self.lines and s.end_lineno not in self.placed:
                    self.placed.add(s.end_lineno)
                    out.append(ast.Expr(ast.Call(ast.Name('__probe__', ast.Load()), [ast.Constant(s.end_lineno)], [])))
            setattr(node, field, out)
        return node


class RefScript(object):
    """A reference script as AST: top level (imports, defs) -> a module namespace; the `__main__` body (or a function's body) in slices."""

    def __init__(self, relpath, function=None):
        """function=None: the statements of the `if __name__ == "__main__":` block; function='main': the body of that top-level def (demo.py)."""
        self.path = os.path.join(REF, relpath)
        tree = ast.parse(open(self.path).read(), self.path)
        self.top = [n for n in tree.body if not _is_main_guard(n)]
        if function is None:
            self.main = [n for n in tree.body if _is_main_guard(n)][0].body
        else:
            self.main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == function][0].body

    def namespace(self):
        ns = {'__name__': 'reference_script', '__file__': self.path}
        exec(compile(ast.Module(self.top, []), self.path, 'exec'), ns)
        return ns

    def run(self, ns, lo, hi, probes=None):
        """Execute the main-body statements of source lines lo..hi in `ns`; `probes` = {line: fn(ns)} called after the
        statement ending on that line."""
        nodes = copy.deepcopy(_slice(self.main, lo, hi))
        assert nodes and nodes[0].lineno == lo and nodes[-1].end_lineno == hi, (lo, hi, [(n.lineno, n.end_lineno) for n in nodes])
        mod = ast.Module(nodes, [])
        if probes:
            tr = _Probes(probes)
            mod = tr.visit(mod)
            assert tr.placed == set(probes), (sorted(tr.placed), sorted(probes))
            ast.fix_missing_locations(mod)
            ns['__probe__'] = lambda line: probes[line](ns)
        exec(compile(mod, self.path, 'exec'), ns)


def verts_digest(v):
    """[B,6890,3] -> strided subsample + per-person checksums (keeps fixtures small)."""
    v = np.asarray(v, dtype=np.float64)
    return {'verts_sub': v[:, ::53].astype(np.float32),
            'verts_sum': v.sum(axis=1).astype(np.float64),
            'verts_l2': np.sqrt((v * v).sum(axis=(1, 2))).astype(np.float64)}


def load_synth(model, sd_np):
    """Overlay the synthetic weights on a reference module's state dict (strict)."""
    sd = model.state_dict()
    missing = [k for k in sd_np if k not in sd]
    assert not missing, missing
    for k in sd:
        if k in sd_np:
            assert tuple(sd[k].shape) == sd_np[k].shape, (k, sd[k].shape, sd_np[k].shape)
            sd[k] = torch.from_numpy(sd_np[k])
    model.load_state_dict(sd, strict=True)
    return model


def run_case(T_mod, P_mod, name, L, H, B, T, use_jreg, seed_w=0, seed_x=1234):
    model = load_synth(T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval(), synth.synthetic_state_dict(L, H, seed_w))
    x = torch.from_numpy(synth.synthetic_windows(B, T, seed_x))
    J = torch.from_numpy(SMPL_NP['J_regressor_h36m']) if use_jreg else None
    seen = []
    real = P_mod.rot6d_to_rotmat
    P_mod.rot6d_to_rotmat = lambda p: (seen.append(p.detach().clone()), real(p))[1]      # the FC loop's result as Regressor.forward hands it on (spin.py:260)
    try:
        with torch.no_grad():
            feat = model.encoder(x)
            out = model(x, J_regressor=J)[0]
            feat_tr = model.encoder(x, is_train=True)
    finally:
        P_mod.rot6d_to_rotmat = real
    d = {'meta': np.array([L, H, B, T, int(use_jreg), seed_w, seed_x], dtype=np.int64),
         'feature': feat.numpy(), 'feature_train': feat_tr.numpy(),
         'theta': out['theta'].numpy(), 'kp_2d': out['kp_2d'].numpy(),
         'kp_3d': out['kp_3d'].numpy(), 'rotmat': out['rotmat'].numpy()}
    d.update(verts_digest(out['verts'].numpy()))
    assert len(seen) == 1
    d['pose6d'] = seen[0].numpy()
    save(name, **d)
    print('wrote', name, {k: v.shape for k, v in d.items()})


def regressor_init_case(P_mod, name, N, n_iter, use_jreg, seed_w=3):
    """The reference's Regressor.forward with per-call init_pose / init_shape / init_cam and a non-default n_iter
    (lib/models/spin.py:240-251): the API the reference offers beyond what TePose.forward uses."""
    reg = P_mod.Regressor().eval()
    sd_np = synth.synthetic_state_dict(1, 64, seed_w)
    load_synth(reg, {k[len('regressor.'):]: v for k, v in sd_np.items() if k.startswith('regressor.') and k[len('regressor.'):] in reg.state_dict()})
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
    save(name, **d)
    print('wrote', name, {k: v.shape for k, v in d.items()})


# ---- flows: the reference's own statements ----------------------------------------------------------------------------
class _Pbar(object):
    """tqdm as evaluate.py:213-214,457 uses it: iterates the clip names, counts, and notes which clips reach the end of the
    loop body (`pbar.set_description`, the last statement of an evaluated clip)."""
    last = None

    def __init__(self, it):
        self.items, self.i, self.done = list(it), -1, []
        _Pbar.last = self

    def __iter__(self):
        for self.i, x in enumerate(self.items):
            yield x

    def set_description(self, s):
        self.done.append(self.i)


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


def run_evaluate_script(EV, T_mod, dataset, L, H, T, seed_w, db, pse, title='repr_wpw_3dpw_model', avg_filter=False, seq=''):
    """evaluate.py's `__main__` body on a synthetic database.  Executed from the file: lines 64-86 (options), 109-137
    (J_regressor, TePose from cfg + checkpoint file, SMPL swap), 141-166 (data paths), 168-462 (keyed clips, VIBE
    bootstrap, window loop, conversion / valid_i / pelvis / metrics, means).  NOT executed: 88-107 (a hard-coded 2 x 1024
    VIBE and a checkpoint download: `model_vibe` is the reference's VIBE class built here at the case's size) and the
    value of `seqlen`, which line 141 fixes at 6 (the fixtures also cover T = 4, 5).  Returns (ns, per-clip records,
    model-call records)."""
    import lib.models.vibe as V_mod
    tmp = tempfile.mkdtemp(prefix='tepose_golden_')
    ckpt = os.path.join(tmp, 'model_best.pth.tar')
    proto = load_synth(T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained=''), synth.synthetic_state_dict(L, H, seed_w))
    torch.save({'performance': 0.0, 'gen_state_dict': proto.state_dict()}, ckpt)
    cfg = CN(DEVICE='cpu', TITLE=title, MODEL=CN(TGRU=CN(NUM_LAYERS=L, HIDDEN_SIZE=H)), DATASET=CN(SEQLEN=T),
             TRAIN=CN(BATCH_SIZE=32, PRETRAINED=ckpt, PRETRAINED_REGRESSOR=''))
    args = types.SimpleNamespace(dataset=dataset, seq=seq, render=False, render_plain=False, frame=0, plot=False, filter=bool(avg_filter))
    if avg_filter:
        _numpy1_for_transformations()
    n = len(db['vid_name'])
    db = dict(db, img_name=np.array(['frame_%06d.jpg' % i for i in range(n)]), bbox=np.zeros((n, 4), dtype=np.float32))
    ns = EV.namespace()
    ns['parse_args'] = lambda: (cfg, 'synthetic.yaml', args)
    ns['joblib'] = types.SimpleNamespace(load=lambda p: db if str(p).endswith('_db.pt') else np.array(pse, copy=True))
    ns['tqdm'] = _Pbar
    ns['model_vibe'] = load_synth(V_mod.VIBE(seqlen=T, n_layers=L, hidden_size=H, add_linear=True, bidirectional=False,
                                             use_residual=True, pretrained='').eval(), synth.synthetic_vibe_state_dict(L, H, seed_w + 1))
    cwd = os.getcwd()
    os.chdir(tmp)                                              # line 144 creates ./output/<dataset>_test_output
    try:
        EV.run(ns, 64, 86)
        EV.run(ns, 109, 137)
        calls = []
        ns['model'].register_forward_hook(lambda m, i, o: calls.append({k: o[-1][k].detach().clone() for k in ('theta', 'kp_3d', 'verts')}))
        EV.run(ns, 141, 166)
        ns['seqlen'] = T
        rec, cur = {}, {}

        def at(line):
            def f(ns):
                ci = _Pbar.last.i
                if line in (291, 294):                          # 294: pred_j3ds = np.vstack(pred_j3ds), raw predictions, bootstrap rows first; 291: the --filter
                    cur.clear()                                 # branch's joints of the re-posed mesh instead (only one of the two lines runs)
                    cur['raw_pred'] = np.array(ns['pred_j3ds'], dtype=np.float32, copy=True)
                elif line == 418:                               # valid_map as the pose metrics use it
                    cur['pose_map'] = np.array(ns['valid_map'], dtype=np.int64, copy=True)
                elif line == 437:                               # the unfiltered per-frame values, checked against the script's filtered ones
                    p, t, s = ns['pred_j3ds'], ns['target_j3ds'], ns['S1_hat']
                    cur['mpjpe_all'] = torch.sqrt(((p - t) ** 2).sum(dim=-1)).numpy().mean(axis=-1) * 1000
                    cur['pa_all'] = torch.sqrt(((s - t) ** 2).sum(dim=-1)).numpy().mean(axis=-1) * 1000
                    assert np.array_equal(cur['mpjpe_all'][cur['pose_map']], ns['mpjpe'])
                    assert np.array_equal(cur['pa_all'][cur['pose_map']], ns['mpjpe_pa'])
                    cur['mpvpe'] = np.array(ns['mpvpe'], copy=True)
                elif line == 442:                               # accel_err over all frames, ends zero
                    cur['accel_all'] = np.array(ns['accel_err'], copy=True)
                elif line == 457:                               # end of an evaluated clip
                    cur['has_accel'] = int(len(cur['pose_map']) > 1)
                    cur['accel_map'] = np.array(ns['valid_map'], dtype=np.int64, copy=True) if cur['has_accel'] else np.zeros(0, dtype=np.int64)
                    if cur['has_accel']:
                        assert np.array_equal(cur['accel_all'][cur['accel_map']], ns['accel_err'])
                    rec[ci] = dict(cur)
            return f

        EV.run(ns, 168, 462, probes={l: at(l) for l in (291, 294, 418, 437, 442, 457)})
    finally:
        os.chdir(cwd)
    assert sorted(rec) == _Pbar.last.done
    return ns, rec, calls


def eval_case(EV, T_mod, name, dataset, L, H, T, lens, seed_w, seed_db, joints=49, invalid_frames=(), avg_filter=False):
    """The reference's evaluation flow end to end on a synthetic `*_db.pt` (run_evaluate_script).  `dataset` is
    `args.dataset`: 'mpii3d' -> data_path '..mpii3d_val_scale12_db.pt' (49 -> 17 joints, valid_i, pelvis -3, no
    J_regressor), 'h36m', '3dpw' (+ MPVPE)."""
    from tepose_amd.data import synthetic_eval_db
    db, pse = synthetic_eval_db(list(lens), seed=seed_db, joints=joints)
    for f in invalid_frames:                                                 # the db's own `valid` column
        db['valid'][f] = 0
    if dataset == 'mpii3d':
        db['valid_i'] = np.concatenate(eval_valid_i(name, lens), axis=0)
    title = 'repr_wpw_h36m_mpii3d_model' if dataset == 'h36m' else 'repr_wpw_3dpw_model'
    ns, per_clip, _ = run_evaluate_script(EV, T_mod, dataset, L, H, T, seed_w, db, pse, title, avg_filter=avg_filter)
    final = {k: float(v) for k, v in ns['full_res'].items()}
    d = {'meta': np.array([L, H, T, seed_w, seed_db, joints] + list(lens), dtype=np.int64),
         'invalid_frames': np.array(list(invalid_frames), dtype=np.int64), 'avg_filter': np.array(int(bool(avg_filter))),
         'tot_num_pose': np.array(ns['tot_num_pose']), 'evaluated_clips': np.array(sorted(per_clip), dtype=np.int64),
         'final_keys': np.array(sorted(final)), 'final_values': np.array([final[k] for k in sorted(final)])}
    if dataset == 'mpii3d':
        d['valid_i'] = db['valid_i']
    for ci, r in per_clip.items():
        for k, v in r.items():
            d['clip%d_%s' % (ci, k)] = np.asarray(v)
    save(name, **d)
    print('wrote', name, final, 'poses', ns['tot_num_pose'], 'clips', sorted(per_clip))


def driver_case(EV, T_mod, name, L, H, N, T, seed_w, seed_x):
    """One clip through the reference's autoregressive loop (evaluate.py:214-269, executed from the file by
    run_evaluate_script); what each `model(...)` call of line 255 returned is recorded by a forward hook."""
    w = synth.synthetic_windows(1, N, seed_x)[0]           # [N,2133]: features + plausible thetas
    db = {'vid_name': np.array(['clip'] * N), 'features': w[:, :2048].copy(), 'joints3D': np.zeros((N, 14, 3), dtype=np.float32),
          'pose': np.zeros((N, 72), dtype=np.float32), 'shape': np.zeros((N, 10), dtype=np.float32), 'valid': np.ones(N, dtype=np.float32)}
    ns, _, calls = run_evaluate_script(EV, T_mod, '3dpw', L, H, T, seed_w, db, w[:, 2048:].copy())
    assert len(calls) == N - T + 1
    theta0 = np.array(w[:T - 1, 2048:], copy=True)
    theta0[:, :3] = [1., 0., 0.]                           # what evaluate.py:177-178 make of the pseudo-theta file
    save(name, meta=np.array([L, H, N, T, seed_w, seed_x], dtype=np.int64), theta_init=theta0,
         theta=np.concatenate([c['theta'].view(-1, 85).numpy() for c in calls]),
         kp_3d=np.concatenate([c['kp_3d'].view(-1, 14, 3).numpy() for c in calls]),
         verts_sub=np.concatenate([c['verts'].view(-1, 6890, 3)[:, ::53].numpy() for c in calls]))
    print('wrote', name, len(calls))


def demo_case(DM, T_mod, name, L, H, N, T, seed_w, seed_x):
    """The live path of demo.py for one tracked person (BASELINE config 5): the `with torch.no_grad():` block of `main` (demo.py:209-262) executed from the
    file -- VIBE over the tracklet's features, its first seq_len - 1 predictions as the theta history, then the sliding window of seq_len frames with theta
    feedback (no J_regressor: 49 joints) -- in a namespace that holds what the lines above it would have left there: `dataloader` (one batch [1, N, 2048] of
    pre-extracted features, demo.py:200-208), `has_keypoints`, `device`, the two models (the reference's classes at the case's size; demo.py:104-155 hard-codes
    2 x 1024 and checkpoint paths), `seq_len`, `num_frames`."""
    import lib.models.vibe as V_mod
    model = load_synth(T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval(), synth.synthetic_state_dict(L, H, seed_w))
    model_vibe = load_synth(V_mod.VIBE(seqlen=N, n_layers=L, hidden_size=H, add_linear=True, bidirectional=False, use_residual=True,
                                       pretrained='').eval(), synth.synthetic_vibe_state_dict(L, H, seed_w + 1))
    feats = torch.from_numpy(synth.synthetic_windows(1, N, seed_x)[:, :, :2048].copy())
    ns = {'torch': torch, 'dataloader': [feats], 'has_keypoints': False, 'device': torch.device('cpu'), 'model': model, 'model_vibe': model_vibe,
          'seq_len': T, 'num_frames': N}
    DM.run(ns, 209, 262)
    theta = torch.cat([ns['pred_cam'], ns['pred_pose'], ns['pred_betas']], dim=1)
    assert theta.shape == (N, 85) and ns['pred_joints3d'].shape == (N, 49, 3) and ns['smpl_joints2d'].shape == (N, 49, 2)
    save(name, meta=np.array([L, H, N, T, seed_w, seed_x], dtype=np.int64), theta=theta.numpy(), kp_3d=ns['pred_joints3d'].numpy(),
         kp_2d=ns['smpl_joints2d'].numpy(), verts_sub=ns['pred_verts'].numpy()[:, ::53])
    print('wrote', name, theta.shape)


def padded_case(T_mod, name, L, H, lens, T, seed_w, seed_x):
    """The batched whole-clip validation loop: the reference's unbound `Trainer.validate` and `Trainer.evaluate`
    (lib/core/trainer.py:294-360, 437-503) called on a namespace object that carries what they read -- the reference
    TePose as `generator`, one padded batch as the validation Datasets emit it (lib/dataset/threedpw_test.py:62-134: clips
    zero-padded to the longest one, arrays staged in float16, theta_pseu with cam = [1, 0, 0], `vidlen_each`, `index`) as
    `valid_loader`.  Every clip advances through ALL vidlen - T + 1 windows, padding included."""
    import lib.core.trainer as TR
    model = load_synth(T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval(), synth.synthetic_state_dict(L, H, seed_w))
    C, vidlen = len(lens), max(lens)
    feats = np.zeros((C, vidlen, 2048), dtype=np.float16)
    theta_pseu = np.zeros((C, vidlen, 85), dtype=np.float16)
    theta_gt = np.zeros((C, vidlen, 85), dtype=np.float16)
    for c, n in enumerate(lens):
        w = synth.synthetic_windows(1, n, seed_x + c)[0]
        feats[c, :n] = w[:, :2048]
        theta_pseu[c, :n] = np.concatenate([np.tile(np.array([1., 0., 0.], dtype=np.float32), (n, 1)), w[:, 2051:]], axis=1)
        g = synth.synthetic_windows(1, n + 1, seed_x + 50 + c)[0, :n]
        theta_gt[c, :n] = np.concatenate([np.tile(np.array([1., 0., 0.], dtype=np.float32), (n, 1)), g[:, 2051:]], axis=1)
    kp3d = np.zeros((C, vidlen, 14, 3), dtype=np.float16)
    for c, n in enumerate(lens):
        kp3d[c, :n] = synth.normal('pad/kp%d_%d' % (seed_x, c), (n, 14, 3), std=0.3)
    target = {'features': torch.from_numpy(feats).float(), 'theta_pseu': torch.from_numpy(theta_pseu).float(),
              'theta': torch.from_numpy(theta_gt).float(), 'kp_2d': torch.zeros((C, vidlen, 14, 3)),
              'vidlen_each': torch.tensor(lens).float().view(C, 1), 'index': torch.arange(C).float().view(C, 1),
              'kp_3d': torch.from_numpy(kp3d).float()}
    calls = []
    model.register_forward_hook(lambda m, i, o: calls.append({k: o[-1][k].detach().clone() for k in ('theta', 'kp_3d', 'verts')}))
    me = types.SimpleNamespace(generator=model, valid_loader=[target], device='cpu', seqlen=T, epoch=0,
                               evaluation_accumulators=dict.fromkeys(['pred_j3d', 'target_j3d', 'target_theta', 'pred_verts',
                                                                      'pred_j3d_tsr', 'target_j3d_tsr', 'vidlen_each']),
                               writer=types.SimpleNamespace(add_scalar=lambda *a, **k: None))
    TR.Trainer.validate(me)
    acc = me.evaluation_accumulators
    assert len(calls) == vidlen - T + 1 == len(acc['pred_j3d'])
    # which rows of each step the trainer kept: read off its accumulators (rows keep their order), not re-derived
    th, vs = [], []
    for c, kept in zip(calls, acc['pred_j3d']):
        rows, r = [], 0
        for k in range(kept.shape[0]):
            while not torch.equal(c['kp_3d'].view(-1, 14, 3)[r], kept[k]):
                r += 1
            rows.append(r)
            r += 1
        th.append(c['theta'].view(-1, 85)[rows].numpy().copy())
        vs.append(c['verts'].view(-1, 6890, 3)[rows][:, ::53].numpy().copy())
    pred_j3d = torch.cat(acc['pred_j3d'], dim=0).numpy().copy()
    pred_j3d_tsr = torch.cat(acc['pred_j3d_tsr'], dim=0).numpy().copy()
    # Trainer.evaluate works in place on the accumulators and returns PA-MPJPE only: take the rest from writer.add_scalar
    scal = {}
    me.writer = types.SimpleNamespace(add_scalar=lambda k, v, global_step=None: scal.__setitem__(k, float(v)))
    pa = TR.Trainer.evaluate(me)
    assert pa == scal['error/pa-mpjpe']
    evald = np.array([scal['error/mpjpe'], scal['error/pa-mpjpe'], scal['error/accel'], scal['error/accel_err']], dtype=np.float64)
    save(name, kp_3d=kp3d, eval_mpjpe_pa_accel_accelerr=evald, eval_pve=np.array(scal['error/pve']), theta=theta_gt,
         meta=np.array([L, H, T, seed_w, seed_x] + list(lens), dtype=np.int64),
         features=feats, theta_pseu=theta_pseu, pred_j3d=pred_j3d,
         pred_theta=np.concatenate(th), pred_verts_sub=np.concatenate(vs), pred_j3d_tsr=pred_j3d_tsr)
    print('wrote', name, pred_j3d.shape, pred_j3d_tsr.shape, scal)


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
            save(name, meta=meta, feature=feat.numpy())
            print('wrote', name, feat.shape, '(encoder only)')
            return
        out = model(x, J_regressor=J)[-1]
    save(name, meta=meta[:6] if (not bidirectional and add_linear and use_residual) else meta, feature=feat.numpy(),
                        theta=out['theta'].numpy(), kp_3d=out['kp_3d'].numpy(), rotmat=out['rotmat'].numpy(),
                        verts_sub=out['verts'].numpy()[:, :, ::53])
    print('wrote', name, feat.shape, out['theta'].shape)


def _affine_from_3_points(src, dst):
    """cv2.getAffineTransform for the stand-in cv2 (the image has no OpenCV): the 2 x 3 matrix that maps three points onto three points.  Only
    lib/data_utils/_img_utils.py's 2-D keypoint cropping reaches it -- a field the hot path never reads and no fixture stores."""
    A = np.concatenate([np.asarray(src, dtype=np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, np.asarray(dst, dtype=np.float64)).T


def reference_validation_loader(which, lens, T, seed_db):
    """(DataLoader, its one batch, joints per frame of the database) of a reference validation Dataset on a synthetic database file:
    which = '3dpw' -> lib/dataset/threedpw_test.py ThreeDPW_TEST on 3dpw_test_*; 'h36m' -> lib/dataset/h36m_val.py Human36M_VAL on
    h36m_test_front_25fps_tight_* (49 'spin' joints); '3dpw_val' -> lib/dataset/threedpw.py ThreeDPW(set='val') = Dataset3D (lib/dataset/dataset_3d.py), the
    class five of the six shipped configs name as TRAIN.DATASET_EVAL (lib/dataset/loaders.py:118), on 3dpw_val_*."""
    import joblib
    import lib.data_utils._img_utils as IU
    from tepose_amd.data import synthetic_eval_db
    if not hasattr(np, 'float'):
        np.float = float                                   # dataset_3d.py:272 still says np.float (removed in NumPy 1.24)
    if which == 'h36m':
        import lib.dataset.h36m_val as DS
        make, stem, nj_db = (lambda: DS.Human36M_VAL(load_opt=None, set='test', seqlen=T, vidlen=max(lens), debug=False)), 'h36m_test_front_25fps_tight', 49
    elif which in ('h36m_ds3d', 'mpii3d_ds3d'):
        # Dataset3D's own H3.6M / MPI-INF-3DHP validation forms (TRAIN.DATASET_EVAL = 'Human36M' / 'MPII3D': no shipped config, but the class offers them):
        # zero ground-truth pose / shape, 14 common resp. 17 mpii3d_test joints
        import lib.dataset.dataset_3d as DS
        if which == 'h36m_ds3d':
            import lib.dataset.h36m as HD
            make, stem, nj_db = (lambda: HD.Human36M(load_opt='repr_wpw_3dpw_model', set='val', seqlen=T, vidlen=max(lens), debug=False)), 'h36m_val', 49
        else:
            import lib.dataset.mpii3d as MD
            make, stem, nj_db = (lambda: MD.MPII3D(load_opt='repr_wpw_3dpw_model', set='val', seqlen=T, vidlen=max(lens), debug=False)), 'mpii3d_val_scale12', 49
    elif which == '3dpw_val':
        import lib.dataset.dataset_3d as DS
        import lib.dataset.threedpw as TD
        make, stem, nj_db = (lambda: TD.ThreeDPW(load_opt='repr_wpw_3dpw_model', set='val', seqlen=T, vidlen=max(lens), overlap=(T - 1) / float(T), debug=False)), '3dpw_val', 14
    else:
        import lib.dataset.threedpw_test as DS
        make, stem, nj_db = (lambda: DS.ThreeDPW_TEST(load_opt=None, set='test', seqlen=T, vidlen=max(lens), debug=False)), '3dpw_test', 14
    db, pse = synthetic_eval_db(list(lens), seed=seed_db, joints=nj_db)
    n = len(db['vid_name'])
    db = dict(db, joints2D=synth.normal('padds/j2d%d' % seed_db, (n, nj_db, 3), std=40.0) + 112.0, img_name=np.array(['f%06d.jpg' % i for i in range(n)]),
              bbox=np.tile(np.array([112., 112., 180., 180.], dtype=np.float32), (n, 1)), frame_id=np.arange(n), valid_i=np.ones((n, 1), dtype=np.float32))
    tmp = tempfile.mkdtemp(prefix='tepose_golden_')
    joblib.dump(db, os.path.join(tmp, stem + '_db.pt'))
    joblib.dump(np.asarray(pse), os.path.join(tmp, stem + '_pseudotheta.pt'))
    DS.TePose_DB_DIR = tmp
    IU.cv2.getAffineTransform = _affine_from_3_points
    ds = make()
    # (lib/dataset/loaders.py:121-126 builds DataLoader(valid_db, batch_size, shuffle=False) with the default collate, which in the torch of the reference's
    # day zipped the per-item string lists -- instance_id, imgname: one entry per real frame -- down to the shortest; today's refuses ragged lists.  The hot
    # path reads tensors only: collate those)
    from torch.utils.data.dataloader import default_collate
    tensors_only = lambda items: default_collate([{k: v for k, v in it.items() if k not in ('instance_id', 'imgname')} for it in items])
    loader = torch.utils.data.DataLoader(ds, batch_size=len(ds), shuffle=False, num_workers=0, collate_fn=tensors_only)
    return loader, next(iter(loader)), nj_db


def padded_ds_case(T_mod, name, L, H, lens, T, seed_w, seed_db, which='3dpw'):
    """The trainer's validation pass from the database FILE on: `3dpw_test_db.pt` / `3dpw_test_pseudotheta.pt` (synthetic, joblib) -> the reference's
    validation Dataset (lib/dataset/threedpw_test.py ThreeDPW_TEST: videos in order of first appearance, short ones dropped, zero padding to the longest,
    float16 staging; which='h36m': lib/dataset/h36m_val.py Human36M_VAL on `h36m_test_front_25fps_tight_*` with 49-joint `joints3D`, which it converts
    spin -> common) -> torch DataLoader (one batch) -> the unbound Trainer.validate / Trainer.evaluate (lib/core/trainer.py:294-360, 437-503).  The
    fixture keeps the batch the Dataset emitted (hot-path fields) next to the accumulators, so tepose_amd.data.padded_validation_batch is pinned to the
    reference's loader and driver.validate_padded / metrics.trainer_evaluate to its loop."""
    import lib.core.trainer as TR
    loader, batch, nj_db = reference_validation_loader(which, lens, T, seed_db)
    model = load_synth(T_mod.TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='').eval(), synth.synthetic_state_dict(L, H, seed_w))
    calls = []
    model.register_forward_hook(lambda m, i, o: calls.append({k: o[-1][k].detach().clone() for k in ('theta', 'kp_3d', 'verts')}))
    scal = {}
    me = types.SimpleNamespace(generator=model, valid_loader=loader, device='cpu', seqlen=T, epoch=0,
                               evaluation_accumulators=dict.fromkeys(['pred_j3d', 'target_j3d', 'target_theta', 'pred_verts',
                                                                      'pred_j3d_tsr', 'target_j3d_tsr', 'vidlen_each']),
                               writer=types.SimpleNamespace(add_scalar=lambda k, v, global_step=None: scal.__setitem__(k, float(v))))
    TR.Trainer.validate(me)
    acc = me.evaluation_accumulators
    vidlen = batch['features'].shape[1]
    assert len(calls) == vidlen - T + 1 == len(acc['pred_j3d'])
    th, vs = [], []
    for c, kept in zip(calls, acc['pred_j3d']):                   # which rows the trainer kept: read off its accumulators
        rows, r = [], 0
        for k in range(kept.shape[0]):
            while not torch.equal(c['kp_3d'].view(-1, 14, 3)[r], kept[k]):
                r += 1
            rows.append(r)
            r += 1
        th.append(c['theta'].view(-1, 85)[rows].numpy().copy())
        vs.append(c['verts'].view(-1, 6890, 3)[rows][:, ::53].numpy().copy())
    pred_j3d = torch.cat(acc['pred_j3d'], dim=0).numpy().copy()
    pred_j3d_tsr = torch.cat(acc['pred_j3d_tsr'], dim=0).numpy().copy()
    pa = TR.Trainer.evaluate(me)
    assert pa == scal['error/pa-mpjpe']
    kept_lens = [int(v) for v in batch['vidlen_each'].view(-1).tolist()]
    save(name, meta=np.array([L, H, T, seed_w, seed_db] + kept_lens, dtype=np.int64), db_lens=np.array(list(lens), dtype=np.int64), db_joints=np.array(nj_db),
         features=batch['features'].numpy().astype(np.float16), theta=batch['theta'].numpy().astype(np.float16),
         theta_pseu=batch['theta_pseu'].numpy().astype(np.float16), kp_3d=batch['kp_3d'].numpy().astype(np.float16),
         vidlen_each=batch['vidlen_each'].numpy(), index=batch['index'].numpy(),
         eval_mpjpe_pa_accel_accelerr=np.array([scal['error/mpjpe'], scal['error/pa-mpjpe'], scal['error/accel'], scal['error/accel_err']], dtype=np.float64),
         eval_pve=np.array(scal['error/pve']), pred_j3d=pred_j3d, pred_theta=np.concatenate(th), pred_verts_sub=np.concatenate(vs), pred_j3d_tsr=pred_j3d_tsr)
    for k in ('features', 'theta', 'theta_pseu', 'kp_3d'):           # float16 staging: the stored halves ARE the batch
        assert np.array_equal(batch[k].numpy(), batch[k].numpy().astype(np.float16).astype(np.float32)), k
    print('wrote', name, pred_j3d.shape, kept_lens, scal)


class _Numpy1(object):
    """The reference's bundled transformations.py calls numpy.array(x, copy=False), which meant "copy only if needed" in the NumPy 1.x it was written
    for and raises in NumPy 2: give that module (only) the old meaning."""

    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def array(obj, *a, **k):
        if k.get('copy', True) is False:
            k.pop('copy')
            return np.asarray(obj, *a, **k)
        return np.array(obj, *a, **k)


def _numpy1_for_transformations():
    import lib.utils.slerp_filter_utils as SF
    SF.numpy = _Numpy1()


def metrics_case(EV):
    """The per-frame metric statements of evaluate.py:413-442 (tensor conversion, pelvis, MPVPE, MPJPE, PA-MPJPE, accel
    error) executed from the file on given prediction / target arrays; data_path selects the pelvis rule (line 420).
    The joint re-orderings come from the reference's `convert_kps`."""
    from lib.data_utils._kp_utils import convert_kps
    out = {}
    for tag, J, path in (('lsp14', 14, 'h36m_test_db.pt'), ('mpii17', 17, 'mpii3d_val_scale12_db.pt')):
        target = torch.from_numpy(synth.normal('met/t' + tag, (60, J, 3), std=0.35))
        pred = target + torch.from_numpy(synth.normal('met/n' + tag, (60, J, 3), std=0.04))
        pred[:, :, 1] *= 1.07                                   # scale + rotation the alignment must remove
        ns = EV.namespace()
        ns.update(pred_j3ds=pred.numpy().copy(), target_j3ds=target.numpy().copy(), valid_map=np.arange(60), tot_num_pose=0,
                  seq_name=tag, data_path=path, plot=False, target_theta=np.zeros((60, 85), dtype=np.float32),
                  pred_verts=np.zeros((60, 6890, 3), dtype=np.float32))
        EV.run(ns, 413, 442)
        out.update({tag + '_pred': pred.numpy(), tag + '_target': target.numpy(), tag + '_mpjpe': ns['mpjpe'],
                    tag + '_pa': ns['mpjpe_pa'], tag + '_accel': ns['accel_err']})
    code = np.arange(49, dtype=np.float64)[None, :, None].repeat(3, axis=2)
    out['spin_to_common'] = convert_kps(code, src='spin', dst='common')[0, :, 0].astype(np.int64)
    out['spin_to_mpii3d_test'] = convert_kps(code, src='spin', dst='mpii3d_test')[0, :, 0].astype(np.int64)
    save('metrics', **out)
    print('wrote metrics', out['spin_to_common'], out['spin_to_mpii3d_test'])


def filter_cases(EV):
    """The reference's `smooth_pose` (lib/utils/smooth_pose.py:24-68: OneEuroFilter over axis-angle poses) and the
    script's own `smooth_pose_mat` (evaluate.py:47-59, slerp over quaternions), both called, not restated."""
    from lib.utils.smooth_pose import smooth_pose
    import lib.utils.slerp_filter_utils as SF

    _numpy1_for_transformations()
    pose = (synth.normal('flt/pose', (50, 24, 3), std=0.4) +
            0.3 * np.sin(np.arange(50, dtype=np.float32) / 6.0)[:, None, None]).astype(np.float32)
    betas = synth.normal('flt/betas', (50, 10), std=0.5)
    sm_verts, hat, sm_joints = smooth_pose(pose, betas, min_cutoff=0.004, beta=0.7)     # (vertices, filtered pose, the wrapper's 49 joints) of every frame
    R = O.batch_rodrigues(torch.from_numpy(pose.reshape(-1, 3))).view(50, 24, 3, 3).numpy().astype(np.float32)
    R[7] = R[7] + synth.normal('flt/noise', (24, 3, 3), std=1e-4)          # not exactly orthonormal
    R_smooth = EV.namespace()['smooth_pose_mat'](R.copy(), ratio=0.3)
    save('filters', pose=pose, pose_hat=hat, R=R, R_smooth=R_smooth, smooth_verts_sub=sm_verts[:, ::53].astype(np.float32), smooth_joints=sm_joints.astype(np.float32))
    print('wrote filters', hat.shape, R_smooth.shape)


def geometry_cases(G):
    """Edge vectors for R->aa (each quaternion branch, angle 0, angles near pi about each axis), rot6d->R (incl.
    degenerate input), and axis-angle->R through the reference's own `batch_rodrigues` (geometry.py:22-65: the only
    Rodrigues the reference HOLDS; quaternion form, |theta + 1e-8|): angles 0, 1e-9, 1e-4, random, pi -+ 1e-3, beyond
    2 pi, both signs."""
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
    vs = [np.zeros(3)]
    for ax in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, -2, 3], [-1, 0.2, 0.1]):
        a = np.asarray(ax, dtype=np.float64) / np.linalg.norm(ax)
        for ang in (1e-9, 1e-4, 0.3, 1.5, 3.0, np.pi - 1e-3, np.pi, np.pi + 1e-3, 2 * np.pi + 0.4, 7.5):
            vs += [a * ang, -a * ang]
    rng = np.random.RandomState(11)
    for _ in range(167):
        v = rng.randn(3)
        vs.append(v / np.linalg.norm(v) * rng.uniform(0, 2 * np.pi))
    rod_in = torch.from_numpy(np.stack(vs).astype(np.float32))          # 288 rows = 12 persons x 24 joints
    assert rod_in.shape[0] == 288
    rod_R = G.batch_rodrigues(rod_in.clone()).view(-1, 3, 3)
    rod_R64 = G.batch_rodrigues(rod_in.double()).view(-1, 3, 3)         # the same statements in float64
    save('geometry', R=R.numpy(), aa=aa.numpy(), x6=x6.numpy(), R6=R6.numpy(), rod_aa=rod_in.numpy(), rod_R=rod_R.numpy(),
         rod_R64=rod_R64.numpy())
    print('wrote geometry', R.shape, aa.shape, R6.shape, rod_R.shape)


def main():
    global OUT
    argv = sys.argv[1:]
    if '--out' in argv:
        OUT = os.path.abspath(argv[argv.index('--out') + 1])
        os.makedirs(OUT, exist_ok=True)
        del argv[argv.index('--out'):argv.index('--out') + 2]
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
    EV = RefScript('evaluate.py')
    DM = RefScript('demo.py', function='main')
    if not argv or 'eval' in argv or 'flows' in argv:
        # BASELINE config 4's three evaluation sets (evaluate.py:394-457 branches)
        eval_case(EV, T_mod, 'eval_mpii3d_L1H64_T5', 'mpii3d', 1, 64, 5, [13, 9, 8, 11, 12, 4, 10], 21, 31)
        eval_case(EV, T_mod, 'eval_h36m_L1H64_T5', 'h36m', 1, 64, 5, [9, 3, 12], 22, 32, invalid_frames=(2, 15, 23))
        eval_case(EV, T_mod, 'eval_h36m14_L1H64_T4', 'h36m', 1, 64, 4, [7, 10], 23, 33, joints=14)
        eval_case(EV, T_mod, 'eval_3dpw_L2H64_T6', '3dpw', 2, 64, 6, [10, 14, 6], 24, 34, invalid_frames=(0, 11))
        # evaluate.py --filter (lines 273-291): slerp-smoothed rotations -> SMPL -> H36M joints of that mesh
        eval_case(EV, T_mod, 'eval_3dpw_filter_L2H64_T6', '3dpw', 2, 64, 6, [10, 14, 6], 24, 34, invalid_frames=(0, 11), avg_filter=True)
        eval_case(EV, T_mod, 'eval_h36m14_filter_L1H64_T4', 'h36m', 1, 64, 4, [7, 10], 23, 33, joints=14, avg_filter=True)
    if not argv or 'flows' in argv:
        driver_case(EV, T_mod, 'driver_L2H128_N40T6', 2, 128, 40, 6, 6, 555)
        driver_case(EV, T_mod, 'driver_L1H64_N9T4', 1, 64, 9, 4, 7, 556)
        padded_case(T_mod, 'padded_L2H128_T5', 2, 128, [23, 9, 17, 5], 5, 14, 700)
        padded_ds_case(T_mod, 'padded_ds_L1H64_T5', 1, 64, [21, 9, 3, 17, 12], 5, 15, 41)
        padded_ds_case(T_mod, 'padded_ds_h36m_L1H64_T4', 1, 64, [8, 22, 2, 11], 4, 16, 42, which='h36m')
        demo_case(DM, T_mod, 'demo_L2H128_N24T6', 2, 128, 24, 6, 16, 811)
        demo_case(DM, T_mod, 'demo_L1H64_N9T8', 1, 64, 9, 8, 17, 812)
        metrics_case(EV)
        filter_cases(EV)
    if argv:
        return
    run_case(T_mod, P_mod, 'tepose_L2H1024_B2T6_j14', 2, 1024, 2, 6, True)
    run_case(T_mod, P_mod, 'tepose_L2H1024_B2T6_j49', 2, 1024, 2, 6, False)
    run_case(T_mod, P_mod, 'tepose_L2H1024_B2T16_j14', 2, 1024, 2, 16, True)
    run_case(T_mod, P_mod, 'tepose_L2H1024_B1T32_j14', 2, 1024, 1, 32, True)
    run_case(T_mod, P_mod, 'tepose_L1H128_B3T5_j49', 1, 128, 3, 5, False, seed_w=3, seed_x=77)
    run_case(T_mod, P_mod, 'tepose_L3H64_B2T4_j14', 3, 64, 2, 4, True, seed_w=4, seed_x=78)
    regressor_init_case(P_mod, 'regressor_init_N5_it2_j14', 5, 2, True)
    regressor_init_case(P_mod, 'regressor_init_N3_it0_j49', 3, 0, False)
    vibe_case('vibe_L2H128_B2N20', 2, 128, 2, 20, 8, 901)
    vibe_case('vibe_L1H64_B1N5', 1, 64, 1, 5, 9, 902)
    vibe_case('vibe_bi_L2H64_B2N7', 2, 64, 2, 7, 10, 903, bidirectional=True, add_linear=False)
    vibe_case('vibe_bi_L1H100_B3N4_nores', 1, 100, 3, 4, 11, 904, bidirectional=True, add_linear=True, use_residual=False)
    vibe_case('vibe_nolin_L1H2048_B1N4', 1, 2048, 1, 4, 12, 905, add_linear=False)
    vibe_case('vibe_nolin_L2H96_B2N6', 2, 96, 2, 6, 13, 906, add_linear=False)
    geometry_cases(G)
    # projection vector
    j = torch.from_numpy(synth.normal('geom/j', (4, 14, 3), std=0.5))
    cam = torch.from_numpy(synth.normal('geom/cam', (4, 3), std=0.1)) + torch.tensor([0.9, 0., 0.])
    save('projection', joints=j.numpy(), cam=cam.numpy(), kp_2d=P_mod.projection(j, cam).numpy())


if __name__ == '__main__':
    main()
