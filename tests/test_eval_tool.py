"""CPU: what tools/evaluate_clips.py decides before it touches a GPU -- the reference's YAML configs
(lib/core/config.py:123-132), the database names of evaluate.py:146-162, the refusal of an incomplete real-data request,
and that real-data runs take EVERY table from the user's files (evaluate.py:109,130-135), never from tepose_amd.synth."""
import glob
import importlib.util
import os

import numpy as np
import pytest

from tepose_amd import synth
from tepose_amd.config import EVAL_SEQLEN, eval_db_paths, get_cfg_defaults, model_kwargs, update_cfg

from _eval_fixture import write_base_data, write_cfg, write_checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONFIGS = '/root/reference/configs'


def _tool():
    spec = importlib.util.spec_from_file_location('evaluate_clips_tool', os.path.join(ROOT, 'tools', 'evaluate_clips.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_and_yaml_merge(tmp_path):
    d = get_cfg_defaults()
    assert (d.MODEL.TGRU.NUM_LAYERS, d.MODEL.TGRU.HIDDEN_SIZE, d.DATASET.SEQLEN) == (1, 2048, 20)    # lib/core/config.py:100,125-126
    write_cfg(tmp_path / 'a.yaml', 'repr_wpw_3dpw_model', 2, 1024, pretrained='data/pretrained_models/x.pth.tar')
    c = update_cfg(str(tmp_path / 'a.yaml'))
    assert model_kwargs(c) == {'n_layers': 2, 'batch_size': 32, 'seqlen': 6, 'hidden_size': 1024,
                               'pretrained': 'data/base_data/spin_model_checkpoint.pth.tar'}
    assert c.TRAIN.PRETRAINED == 'data/pretrained_models/x.pth.tar' and c.TRAIN.MOT_DISCR.OPTIM == 'Adam'
    assert c.TRAIN.MOT_DISCR.LR == 1e-2 and c.LOSS.KP_3D_W == 30.                 # untouched defaults survive the merge
    assert get_cfg_defaults().TITLE == 'default'                                   # the defaults are not mutated
    (tmp_path / 'bad.yaml').write_text("MODEL:\n  TGRU:\n    NUM_LAYER: 2\n")
    with pytest.raises(KeyError):
        update_cfg(str(tmp_path / 'bad.yaml'))
    (tmp_path / 'bad2.yaml').write_text("MODEL:\n  TGRU:\n    NUM_LAYERS: 'two'\n")
    with pytest.raises(ValueError):
        update_cfg(str(tmp_path / 'bad2.yaml'))


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason='build container only: reads the shipped YAML files')
def test_the_six_shipped_configs_merge():
    want = {'config.yaml': ('default', 2, 1024, 16), 'repr_wopw_3dpw_model.yaml': ('repr_wopw_3dpw_model', 2, 1024, 6),
            'repr_wopw_h36m_model.yaml': ('repr_wopw_h36m_model', 2, 1024, 6),
            'repr_wopw_mpii3d_model.yaml': ('repr_wopw_mpii3d_model', 2, 1024, 6),
            'repr_wpw_3dpw_model.yaml': ('repr_wpw_3dpw_model', 2, 1024, 6),
            'repr_wpw_h36m_mpii3d_model.yaml': ('repr_wpw_h36m_mpii3d_model', 2, 1024, 6)}
    files = sorted(glob.glob(os.path.join(REF_CONFIGS, '*.yaml')))
    assert sorted(os.path.basename(f) for f in files) == sorted(want)
    for f in files:
        c = update_cfg(f)
        kw = model_kwargs(c)
        assert (c.TITLE, kw['n_layers'], kw['hidden_size'], kw['seqlen']) == want[os.path.basename(f)]
        assert model_kwargs(c, seqlen=EVAL_SEQLEN)['seqlen'] == 6                  # evaluate.py:141


def test_database_names_follow_evaluate_py():
    c = get_cfg_defaults()
    assert eval_db_paths(c, '3dpw', 'D') == ('D/3dpw_test_db.pt', 'D/3dpw_test_pseudotheta.pt')
    assert eval_db_paths(c, '3dpw', 'D', render=True)[0] == 'D/3dpw_test_all_db.pt'
    assert eval_db_paths(c, 'mpii3d', 'D') == ('D/mpii3d_val_scale12_db.pt', 'D/mpii3d_val_scale12_pseudotheta.pt')
    c.TITLE = 'repr_wpw_h36m_mpii3d_model'
    assert eval_db_paths(c, 'h36m', 'D')[0] == 'D/h36m_test_25fps_nosmpl_db.pt'
    c.TITLE = 'repr_wopw_h36m_model'
    assert eval_db_paths(c, 'h36m', 'D')[1] == 'D/h36m_test_front_25fps_tight_nosmpl_pseudotheta.pt'
    c.TITLE = 'repr_wpw_3dpw_model'
    with pytest.raises(ValueError):
        eval_db_paths(c, 'h36m', 'D')
    with pytest.raises(ValueError):
        eval_db_paths(c, 'coco', 'D')


def test_real_data_requests_are_complete_or_refused(tmp_path):
    import joblib
    from tepose_amd.data import synthetic_eval_db
    T = _tool()
    db, pse = synthetic_eval_db([7, 9], seed=3)
    joblib.dump(db, tmp_path / '3dpw_test_db.pt')
    joblib.dump(pse, tmp_path / '3dpw_test_pseudotheta.pt')
    smpl_np = {k: np.array(v, copy=True) for k, v in synth.synthetic_smpl(0).items()}
    smpl_np['J_regressor_h36m'] = smpl_np['J_regressor_h36m'][::-1].copy()        # NOT the synthetic table: must come from the file
    mean = {k: np.asarray(v) * 0.5 for k, v in synth.synthetic_mean_params(0).items()}
    write_base_data(tmp_path / 'base', smpl_np, mean)
    write_checkpoint(tmp_path / 'tepose.pth.tar', synth.synthetic_state_dict(1, 64, 2), prefix='module.')
    write_checkpoint(tmp_path / 'vibe.pth.tar', synth.synthetic_vibe_state_dict(1, 64, 3))
    write_cfg(tmp_path / 'c.yaml', 'repr_wpw_3dpw_model', 1, 64, pretrained=str(tmp_path / 'tepose.pth.tar'))
    db_args = ['--db', str(tmp_path / '3dpw_test_db.pt'), '--pseudotheta', str(tmp_path / '3dpw_test_pseudotheta.pt')]
    full = db_args + ['--base-data', str(tmp_path / 'base'), '--cfg', str(tmp_path / 'c.yaml'),
                      '--vibe-ckpt', str(tmp_path / 'vibe.pth.tar')]
    # refusals: no base data; no checkpoint; no bootstrap checkpoint; missing database; --db without --pseudotheta
    for argv in (db_args + ['--ckpt', str(tmp_path / 'tepose.pth.tar'), '--vibe-ckpt', str(tmp_path / 'vibe.pth.tar')],
                 db_args + ['--base-data', str(tmp_path / 'base'), '--vibe-ckpt', str(tmp_path / 'vibe.pth.tar')],
                 db_args + ['--base-data', str(tmp_path / 'base'), '--cfg', str(tmp_path / 'c.yaml')],
                 ['--db-dir', str(tmp_path / 'nowhere'), '--base-data', str(tmp_path / 'base'), '--cfg', str(tmp_path / 'c.yaml'),
                  '--vibe-ckpt', str(tmp_path / 'vibe.pth.tar')],
                 ['--db', str(tmp_path / '3dpw_test_db.pt'), '--base-data', str(tmp_path / 'base')]):
        with pytest.raises(SystemExit) as e:
            T.resolve_plan(T.parse_args(argv))
        assert e.value.code not in (0, None)
    plan = T.resolve_plan(T.parse_args(full))
    assert plan['real'] and (plan['layers'], plan['hidden'], plan['seqlen']) == (1, 64, 6)
    assert (plan['vibe_layers'], plan['vibe_hidden']) == (2, 1024)                  # evaluate.py:93-101
    assert plan['ckpt'] == str(tmp_path / 'tepose.pth.tar')                         # from the YAML's TRAIN.PRETRAINED
    # the same request through --db-dir: names derived as evaluate.py:146-148
    plan2 = T.resolve_plan(T.parse_args(['--db-dir', str(tmp_path)] + full[4:]))
    assert plan2['db'] == str(tmp_path / '3dpw_test_db.pt')
    # every table comes from the files
    a = T.load_assets(dict(plan, seq=''))
    assert a['source'] == 'files'
    assert np.array_equal(a['J_regressor_h36m'], smpl_np['J_regressor_h36m'])
    assert not np.array_equal(a['J_regressor_h36m'], synth.synthetic_smpl(0)['J_regressor_h36m'])
    assert np.array_equal(a['mean_tepose']['pose'], mean['pose'].astype(np.float32))
    assert np.array_equal(a['smpl_tables']['posedirs'], smpl_np['posedirs'])
    assert np.array_equal(a['smpl_tables']['J_regressor_extra'], smpl_np['J_regressor_extra'])
    assert [len(c['features']) for c in a['clips'].values()] == [7, 9]
    # a base directory with a file missing names it
    os.remove(tmp_path / 'base' / 'J_regressor_extra.npy')
    with pytest.raises(FileNotFoundError, match='J_regressor_extra.npy'):
        T.load_assets(dict(plan, seq=''))
    # the synthetic harness stays what it was
    p0 = T.resolve_plan(T.parse_args([]))
    assert not p0['real'] and (p0['layers'], p0['hidden'], p0['seqlen'], p0['vibe_layers']) == (2, 1024, 6, 2)


def test_the_tool_source_reads_no_synthetic_table_in_real_mode():
    """VERDICT r4 missing #2: `synth.synthetic_smpl(0)['J_regressor_h36m']` used to be the regressor of every run."""
    src = open(os.path.join(ROOT, 'tools', 'evaluate_clips.py')).read()
    body = src[src.index('def load_assets'):src.index('def main')]
    real = body[body.index("if plan['real']:"):body.index('smpl_np = synth.synthetic_smpl(0)')]
    assert 'synth.' not in real
    main = src[src.index('def main'):]
    assert "synth.synthetic_smpl" not in main and "smpl_np['J_regressor_h36m']" not in main


def test_check_mode_reports_real_data_readiness(tmp_path, capsys):
    """`tools/evaluate_clips.py --check`: found / missing licence-gated files (reference README.md:21-25,37-62) with their shapes, the
    published row the run will be compared with; exit code 0 only when everything a real-data run needs is there.  Fabricated files, CPU only."""
    import joblib
    import json
    from tepose_amd.config import compare_with_published, published_row
    from tepose_amd.data import synthetic_eval_db
    T = _tool()
    db, pse = synthetic_eval_db([7, 9], seed=3)
    joblib.dump(db, tmp_path / '3dpw_test_db.pt')
    joblib.dump(pse, tmp_path / '3dpw_test_pseudotheta.pt')
    smpl_np = synth.synthetic_smpl(0)
    write_base_data(tmp_path / 'base', smpl_np, synth.synthetic_mean_params(0))
    write_checkpoint(tmp_path / 'tepose.pth.tar', synth.synthetic_state_dict(1, 64, 2), prefix='module.')
    write_checkpoint(tmp_path / 'vibe.pth.tar', synth.synthetic_vibe_state_dict(1, 64, 3))
    write_cfg(tmp_path / 'c.yaml', 'repr_wpw_3dpw_model', 1, 64, pretrained=str(tmp_path / 'tepose.pth.tar'))
    argv = ['--check', '--cfg', str(tmp_path / 'c.yaml'), '--dataset', '3dpw', '--base-data', str(tmp_path / 'base'), '--db-dir', str(tmp_path),
            '--vibe-ckpt', str(tmp_path / 'vibe.pth.tar')]
    rep = T.readiness_report(T.parse_args(argv))
    assert rep['ready'] and not rep['missing'] and not rep['unreadable'] and rep['cfg_title'] == 'repr_wpw_3dpw_model'
    assert rep['published'] == {'mpjpe_pa': 52.3, 'mpjpe': 84.6, 'mpvpe': 100.3, 'accel_err': 11.4}
    by = {os.path.basename(str(f['path'])): f for f in rep['files']}
    assert by['J_regressor_h36m.npy']['shape'] == [17, 6890] and by['J_regressor_extra.npy']['shape'] == [9, 6890]
    assert by['smpl_mean_params.npz']['arrays'] == {'pose': [144], 'shape': [10], 'cam': [3]}
    assert by['SMPL_NEUTRAL.pkl']['arrays']['v_template'] == [6890, 3]
    assert by['3dpw_test_db.pt']['clips'] == 2 and by['3dpw_test_db.pt']['frames'] == 16 and by['3dpw_test_db.pt']['arrays']['features'] == [16, 2048]
    assert by['3dpw_test_pseudotheta.pt']['shape'] == [16, 85]
    assert by['tepose.pth.tar']['tensors'] > 20 and abs(by['tepose.pth.tar']['performance'] - 51.2) < 1e-9
    # through main(): one JSON line, exit code 0
    import sys
    old = sys.argv
    sys.argv = ['evaluate_clips.py'] + argv
    try:
        with pytest.raises(SystemExit) as e:
            T.main()
    finally:
        sys.argv = old
    assert e.value.code == 0
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])['ready'] is True
    # something missing / of the wrong shape: named, exit code 1
    os.remove(tmp_path / '3dpw_test_pseudotheta.pt')
    np.save(tmp_path / 'base' / 'J_regressor_extra.npy', np.zeros((9, 100), dtype=np.float32))
    rep = T.readiness_report(T.parse_args(argv))
    assert not rep['ready'] and rep['missing'] == [str(tmp_path / '3dpw_test_pseudotheta.pt')]
    assert rep['unreadable'] == [str(tmp_path / 'base' / 'J_regressor_extra.npy')]
    # nothing on disk at all (this container): every licence-gated default path is listed as missing
    rep = T.readiness_report(T.parse_args(['--check', '--dataset', 'mpii3d', '--base-data', str(tmp_path / 'nowhere'), '--db-dir', str(tmp_path / 'nowhere')]))
    assert not rep['ready'] and len(rep['missing']) >= 7 and any('mpii3d_val_scale12_db.pt' in p for p in rep['missing'] if p)
    # the comparison a real-data run prints
    assert published_row('repr_wopw_h36m_model', 'h36m') == {'mpjpe_pa': 41.2, 'mpjpe': 61.6, 'accel_err': 12.0}
    cmp = compare_with_published({'mpjpe': 84.9, 'mpjpe_pa': 52.3, 'accel_err': 11.0, 'mpvpe': 100.0}, 'repr_wpw_3dpw_model', '3dpw')
    assert abs(cmp['mpjpe']['diff'] - 0.3) < 1e-9 and cmp['mpjpe_pa']['diff'] == 0.0 and cmp['mpvpe']['published'] == 100.3
    assert compare_with_published({'mpjpe': 1.0}, 'some_other_title', '3dpw') is None
