"""The fixture generator pins the flow goldens to the reference's OWN statements (tests/golden/make_golden.py executes
AST slices of /root/reference/evaluate.py and the unbound Trainer.validate / Trainer.evaluate): it must not contain
retyped runs of those files, and re-running it must reproduce the committed flow fixtures bit for bit.

Build container only (skipped where /root/reference or the generator is absent: the generator is listed in
`.gpurunignore` and never travels to the GPU box)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GEN = os.path.join(ROOT, 'tests', 'golden', 'make_golden.py')

pytestmark = pytest.mark.skipif(not (os.path.isfile(os.path.join(REF, 'evaluate.py')) and os.path.isfile(GEN)),
                                reason='reference checkout or generator absent')


def _code_lines(path):
    out = []
    for l in open(path):
        l = re.sub(r'\s+', '', l.split('#')[0])
        if l:
            out.append(l)
    return out


@pytest.mark.parametrize('ref', ['evaluate.py', 'demo.py', 'lib/core/trainer.py', 'lib/core/tester.py', 'lib/utils/smooth_pose.py',
                                 'lib/utils/eval_utils.py'])
def test_generator_holds_no_run_of_reference_lines(ref):
    g, r = _code_lines(GEN), _code_lines(os.path.join(REF, ref))
    runs = {tuple(r[i:i + 3]) for i in range(len(r) - 2)}
    hits = [i for i in range(len(g) - 2) if tuple(g[i:i + 3]) in runs]
    assert not hits, [g[i] for i in hits[:5]]


def test_generator_is_not_in_the_gpu_payload():
    with open(os.path.join(ROOT, '.gpurunignore')) as f:
        assert 'tests/golden/make_golden.py' in [l.strip() for l in f]


def test_flow_fixtures_regenerate_bit_identically(tmp_path):
    p = subprocess.run([sys.executable, GEN, '--out', str(tmp_path), 'flows'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    made = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith('.npz'))
    assert len(made) == 15 and sum(f.startswith('eval_') for f in made) == 6 and sum(f.startswith('demo_') for f in made) == 2, made
    for f in made:
        a, b = np.load(os.path.join(str(tmp_path), f)), np.load(os.path.join(ROOT, 'tests', 'golden', f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == 'f'), (f, k)


def test_database_names_equal_what_evaluate_py_derives():
    """tepose_amd.config.eval_db_paths against lines 141-166 of the reference's evaluate.py, executed for every (config TITLE, --dataset, render) the script has
    a database for -- and the combinations it has none for."""
    import importlib.util
    import tempfile
    import types
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location('make_golden_mod', GEN)
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from tepose_amd.config import eval_db_paths, get_cfg_defaults
    EV = mg.RefScript('evaluate.py')
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())                       # line 144 creates ./output/<dataset>_test_output
    try:
        for title, dataset, render in (('repr_wpw_3dpw_model', '3dpw', False), ('repr_wopw_3dpw_model', '3dpw', True),
                                       ('repr_wpw_h36m_mpii3d_model', 'h36m', False), ('repr_wopw_h36m_model', 'h36m', False),
                                       ('repr_wpw_h36m_mpii3d_model', 'mpii3d', False), ('repr_wopw_mpii3d_model', 'mpii3d', False)):
            import os.path as osp
            from pathlib import Path
            ns = {'osp': osp, 'Path': Path, 'TePose_DB_DIR': 'data/preprocessed_data', 'cfg': types.SimpleNamespace(TITLE=title),
                  'target_dataset': dataset, 'set': 'test', 'render': render, 'print': lambda *a, **k: None}
            EV.run(ns, 141, 166)
            cfg = get_cfg_defaults()
            cfg.TITLE = title
            assert eval_db_paths(cfg, dataset, 'data/preprocessed_data', render=render) == (ns['data_path'], ns['psetheta_file']), (title, dataset)
            assert ns['seqlen'] == 6
    finally:
        os.chdir(cwd)


def test_clip_selection_equals_what_evaluate_py_keys():
    """tepose_amd.data.split_db_into_clips (vid_name grouping in np.unique order, the db's `valid` column, `--seq` substring filter, camera of the
    pseudo-theta forced to [1, 0, 0]) against `dataset_data` as lines 169-207 of the reference's evaluate.py leave it, executed on a synthetic database."""
    import importlib.util
    import numpy as np
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location('make_golden_mod2', GEN)
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.install_stubs()
    import lib.models.tepose as T_mod
    from tepose_amd.data import split_db_into_clips, synthetic_eval_db
    EV = mg.RefScript('evaluate.py')
    db, pse = synthetic_eval_db([7, 9, 8, 12], seed=77)
    db['vid_name'] = np.concatenate([np.array([n] * l) for n, l in zip(['walk_b', 'run_a', 'walk_a', 'sit_c'], [7, 9, 8, 12])])   # not in np.unique order
    for f in (1, 8, 20):
        db['valid'][f] = 0
    for seq in ('', 'walk', 'run_a', 'nothing'):
        ns, per_clip, _ = mg.run_evaluate_script(EV, T_mod, '3dpw', 1, 64, 6, 5, db, pse, seq=seq)
        ours = split_db_into_clips(db, pse, target_action=seq)
        assert list(ours) == list(ns['dataset_data'])
        for k in ours:
            for f in ('features', 'joints3D', 'theta_pseu', 'pose', 'shape'):
                assert np.array_equal(np.asarray(ours[k][f], dtype=np.float64), np.asarray(ns['dataset_data'][k][f], dtype=np.float64)), (seq, k, f)


def test_the_default_validation_dataset_class_emits_the_same_batch():
    """Five of the six shipped configs validate with TRAIN.DATASET_EVAL = 'ThreeDPW' (lib/dataset/loaders.py:118: `ThreeDPW(set='val')`, i.e. Dataset3D in
    lib/dataset/dataset_3d.py), the sixth with 'Human36M_VAL'.  On the same synthetic database the reference's ThreeDPW(set='val') emits the batch its
    ThreeDPW_TEST emits -- the one tests/golden/padded_ds_L1H64_T5.npz holds and tepose_amd.data.padded_validation_batch reproduces bit for bit."""
    import importlib.util
    import numpy as np
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location('make_golden_mod3', GEN)
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.install_stubs()
    from tepose_amd.data import padded_validation_batch, synthetic_eval_db
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'padded_ds_L1H64_T5.npz'))
    T, seed_db, lens = int(g['meta'][2]), int(g['meta'][4]), [int(v) for v in g['db_lens']]
    _, batch, nj = mg.reference_validation_loader('3dpw_val', lens, T, seed_db)
    assert nj == 14
    for k in ('features', 'theta', 'theta_pseu', 'kp_3d'):
        assert np.array_equal(batch[k].numpy(), g[k].astype(np.float32)), k
    assert np.array_equal(batch['vidlen_each'].numpy(), g['vidlen_each']) and np.array_equal(batch['index'].numpy(), g['index'])
    ours = padded_validation_batch(*synthetic_eval_db(lens, seed=seed_db, joints=14), seqlen=T)
    for k in ('features', 'theta', 'theta_pseu', 'kp_3d', 'vidlen_each', 'index'):
        assert np.array_equal(ours[k].numpy(), batch[k].numpy()), k


def test_dataset3d_validation_forms_of_h36m_and_mpii3d():
    """Dataset3D(set='val') on H3.6M / MPI-INF-3DHP (lib/dataset/dataset_3d.py:182-194,214-233: TRAIN.DATASET_EVAL = 'Human36M' / 'MPII3D'): the 14 common resp.
    17 mpii3d_test joints of a 49-joint database, ground-truth pose / shape zero.  tepose_amd.data.padded_validation_batch(eval_class=...) against the
    reference classes' batch on the same synthetic database, bit for bit."""
    import importlib.util
    import numpy as np
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location('make_golden_mod4', GEN)
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.install_stubs()
    from tepose_amd.data import padded_validation_batch, synthetic_eval_db
    lens, T, seed_db = [11, 3, 20, 7], 5, 55
    for which, cls, nj in (('h36m_ds3d', 'Human36M', 14), ('mpii3d_ds3d', 'MPII3D', 17)):
        _, batch, nj_db = mg.reference_validation_loader(which, lens, T, seed_db)
        assert nj_db == 49 and batch['kp_3d'].shape[2] == nj and not batch['theta'][:, :, 3:].any()
        ours = padded_validation_batch(*synthetic_eval_db(lens, seed=seed_db, joints=49), seqlen=T, eval_class=cls)
        for k in ('features', 'theta', 'theta_pseu', 'kp_3d', 'vidlen_each', 'index'):
            assert np.array_equal(ours[k].numpy(), batch[k].numpy()), (which, k)
