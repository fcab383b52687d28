"""GPU: end-to-end clip evaluation (VIBE bootstrap -> autoregressive windows -> metrics ->
records) on a synthetic database in the reference's `*_db.pt` schema, against the same
pipeline assembled from oracle pieces on the CPU."""
import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu


def test_eval_pipeline_matches_oracle_pipeline(tmp_path):
    import joblib
    from oracle import tepose_ref as O
    from tepose_amd.data import load_eval_db, synthetic_eval_db
    from tepose_amd.evaluate import evaluate_clips, gather_and_reduce
    from tepose_amd.metrics import SPIN_TO_COMMON
    from tepose_amd.smpl import SMPL
    from tepose_amd.testing import build_model
    from tepose_amd.vibe import VIBE
    L, H, T = 1, 64, 5
    smpl_np = synth.synthetic_smpl(0)
    db, pse = synthetic_eval_db([11, 3, 8, 14], seed=1)
    joblib.dump(db, tmp_path / '3dpw_test_db.pt')
    joblib.dump(pse, tmp_path / '3dpw_test_pseudotheta.pt')
    clips = load_eval_db(tmp_path / '3dpw_test_db.pt', tmp_path / '3dpw_test_pseudotheta.pt')
    assert list(clips) == ['clip_00', 'clip_01', 'clip_02', 'clip_03']
    assert np.all(clips['clip_00']['theta_pseu'][:, :3] == [1, 0, 0])

    model, state, _ = build_model(L, H, seed=4, device='cuda', smpl_np=smpl_np)
    vstate = synth.synthetic_vibe_state_dict(1, 64, 5)
    mean = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0],
            'cam': vstate['regressor.init_cam'][0]}
    vibe = VIBE(seqlen=T, n_layers=1, hidden_size=64, add_linear=True, use_residual=True, pretrained='',
                smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean)
    sd = vibe.state_dict()
    for k, v in vstate.items():
        sd[k] = torch.from_numpy(v)
    vibe.load_state_dict(sd)
    vibe = vibe.cuda().eval()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    recs, mine = evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset='3dpw')
    assert mine == [3, 0, 2] or sorted(mine) == [0, 2, 3]                 # clip_01 (3 frames) is skipped
    got = gather_and_reduce(recs)

    # the same pipeline from oracle pieces
    mp, pa, ac, mv = [], [], [], []
    for name, c in clips.items():
        n = len(c['features'])
        if n < T:
            continue
        boot = O.vibe_fwd(vstate, smpl_np, c['features'][None, :T], 1, J_regressor=smpl_np['J_regressor_h36m'])
        seq = O.run_clip(state, smpl_np, c['features'], c['theta_pseu'][:T - 1], T, L,
                         J_regressor=smpl_np['J_regressor_h36m'])
        pj = torch.cat([boot['kp_3d'][:T - 1], seq['kp_3d']])
        pv = torch.cat([boot['verts'][:T - 1], seq['verts']])
        tj = torch.from_numpy(c['joints3D'])[:, SPIN_TO_COMMON]
        m = O.joint_metrics(pj, tj)
        tt = np.concatenate([np.zeros((n, 3), np.float32), c['pose'], c['shape']], axis=1)
        gv = O.verts_from_theta(smpl_np, tt)
        mp.append(m['mpjpe']); pa.append(m['pa_mpjpe']); ac.append(m['accel'][1:-1])
        mv.append(torch.sqrt(((pv - gv) ** 2).sum(-1)).mean(-1) * 1000)
    want = {'mpjpe': torch.cat(mp).mean().item(), 'mpjpe_pa': torch.cat(pa).mean().item(),
            'accel_err': torch.cat(ac).mean().item(), 'mpvpe': torch.cat(mv).mean().item()}
    for k in want:
        assert abs(got[k] - want[k]) < 0.02, (k, got[k], want[k])         # mm


def _fixture_models(fx):
    from _eval_fixture import vibe_state
    from tepose_amd.smpl import SMPL
    from tepose_amd.testing import build_model
    from tepose_amd.vibe import VIBE
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(fx['L'], fx['H'], seed=fx['seed_w'], device='cuda', smpl_np=smpl_np, seqlen=fx['T'])
    vstate, mean = vibe_state(fx['L'], fx['H'], fx['seed_w'] + 1)
    vibe = VIBE(seqlen=fx['T'], n_layers=fx['L'], hidden_size=fx['H'], add_linear=True, use_residual=True, pretrained='',
                smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean)
    sd = vibe.state_dict()
    for k, v in vstate.items():
        sd[k] = torch.from_numpy(v)
    vibe.load_state_dict(sd)
    return model, vibe.cuda().eval(), smpl_np


@pytest.mark.parametrize('case', ['eval_mpii3d_L1H64_T5', 'eval_h36m_L1H64_T5', 'eval_h36m14_L1H64_T4', 'eval_3dpw_L2H64_T6',
                                  'eval_3dpw_filter_L2H64_T6', 'eval_h36m14_filter_L1H64_T4'])
def test_dataset_branches_match_the_reference_flow(case):
    """BASELINE config 4's three evaluation sets end to end (VIBE bootstrap -> windows -> joint conversion -> valid_i filter
    -> pelvis -> metrics -> records -> frame-weighted means) against tests/golden/eval_*.npz = the reference's own
    evaluate.py flow run with its TePose / VIBE classes and eval_utils functions: mpii3d with J_regressor=None (49 joints ->
    mpii3d_test, pelvis = joint -3, valid_i holes, a clip without valid frames, one with a single valid frame), h36m with
    49- and 14-joint targets and invalid db frames, 3dpw with MPVPE; `*_filter_*`: evaluate.py --filter (lines 273-291: slerp-smoothed rotations ->
    SMPL -> the H36M joints of that mesh; MPVPE on the unfiltered vertices).  0.01 mm on every per-clip sum and the final means."""
    from _eval_fixture import load
    from tepose_amd.evaluate import clip_metric_record, evaluate_clips, gather_and_reduce
    fx = load(case)
    model, vibe, smpl_np = _fixture_models(fx)
    J = None if fx['dataset'] == 'mpii3d' else torch.from_numpy(smpl_np['J_regressor_h36m'])
    recs, mine = evaluate_clips(model, vibe, fx['clips'], fx['T'], J_regressor=J, dataset=fx['dataset'], avg_filter=fx['avg_filter'])
    recs = recs.cpu()
    assert sorted(int(r[0]) for r in recs) == sorted(fx['per_clip'])            # same clips skipped as the reference
    assert int(recs[:, 1].sum()) == fx['tot_num_pose']
    names = list(fx['clips'])
    for r in recs:
        g = fx['per_clip'][int(r[0])]
        n = len(g['pose_map'])
        assert int(r[1]) == n
        assert abs(float(r[2]) - float(g['mpjpe_all'][g['pose_map']].sum())) < 1e-2 * n
        assert abs(float(r[3]) - float(g['pa_all'][g['pose_map']].sum())) < 1e-2 * n
        assert int(r[4]) == (len(g['accel_map']) if int(g['has_accel']) else 0)
        assert abs(float(r[5]) - float(g['accel_all'][g['accel_map']].sum())) < 1e-2 * max(1, len(g['accel_map']))
        if fx['dataset'] == '3dpw':
            assert int(r[6]) == len(g['mpvpe']) and abs(float(r[7]) - float(g['mpvpe'].sum())) < 1e-2 * len(g['mpvpe'])
        else:
            assert int(r[6]) == 0
    got = gather_and_reduce(recs.cuda())
    assert set(got) == set(fx['final'])
    for k, v in fx['final'].items():
        assert abs(got[k] - v) < 1e-2, (k, got[k], v)
    # the metric block alone on the reference's own raw predictions (no model in between)
    for ci, g in fx['per_clip'].items():
        pv = torch.zeros(len(g['raw_pred']), 6890, 3, device='cuda')
        r = clip_metric_record(model, ci, fx['clips'][names[ci]], torch.from_numpy(g['raw_pred']).cuda(), pv,
                               'h36m' if fx['dataset'] == '3dpw' else fx['dataset'])
        assert abs(float(r[2]) - float(g['mpjpe_all'][g['pose_map']].sum())) < 2e-3 * len(g['pose_map'])
        assert abs(float(r[3]) - float(g['pa_all'][g['pose_map']].sum())) < 2e-3 * len(g['pose_map'])


@pytest.mark.parametrize('case,world', [('eval_mpii3d_L1H64_T5', 1), ('eval_h36m_L1H64_T5', 2), ('eval_3dpw_filter_L2H64_T6', 1)])
def test_the_eval_tool_on_files_alone_reproduces_the_reference_flow(tmp_path, case, world):
    """tools/evaluate_clips.py in real-data mode: database + pseudo-theta (joblib), base data (J_regressor_h36m.npy,
    smpl_mean_params.npz, SMPL_NEUTRAL.pkl, J_regressor_extra.npy), experiment YAML and the two checkpoints all read from
    files -- the JSON line's metrics equal the reference flow's (tests/golden/eval_*.npz) to 0.01 mm; world 2 = two ranks
    sharing cuda:0 over gloo (clip sharding + weight broadcast + record gather)."""
    import json
    import os
    import subprocess
    import sys
    import joblib
    from _eval_fixture import load, vibe_state, write_base_data, write_cfg, write_checkpoint
    from tepose_amd.data import synthetic_eval_db
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = load(case)
    import numpy as np
    g = np.load(os.path.join(root, 'tests', 'golden', case + '.npz'))
    lens = [int(v) for v in g['meta'][6:]]
    db, pse = synthetic_eval_db(lens, seed=int(g['meta'][4]), joints=fx['joints'])
    for f in g['invalid_frames']:
        db['valid'][int(f)] = 0
    stem = {'mpii3d': 'mpii3d_val_scale12', 'h36m': 'h36m_test_25fps_nosmpl', '3dpw': '3dpw_test'}[fx['dataset']]
    title = 'repr_wpw_3dpw_model' if fx['dataset'] == '3dpw' else 'repr_wpw_h36m_mpii3d_model'
    if fx['dataset'] == 'mpii3d':
        db['valid_i'] = g['valid_i']
    joblib.dump(db, tmp_path / (stem + '_db.pt'))
    joblib.dump(pse, tmp_path / (stem + '_pseudotheta.pt'))
    smpl_np = synth.synthetic_smpl(0)
    state = synth.synthetic_state_dict(fx['L'], fx['H'], fx['seed_w'])
    vstate, _ = vibe_state(fx['L'], fx['H'], fx['seed_w'] + 1)
    mean = {'pose': state['regressor.init_pose'][0], 'shape': state['regressor.init_shape'][0], 'cam': state['regressor.init_cam'][0]}
    write_base_data(tmp_path / 'base', smpl_np, mean)
    write_checkpoint(tmp_path / 'tepose.pth.tar', state)
    write_checkpoint(tmp_path / 'vibe.pth.tar', vstate)
    write_cfg(tmp_path / 'c.yaml', title, fx['L'], fx['H'], pretrained=str(tmp_path / 'tepose.pth.tar'))
    cmd = [sys.executable, 'tools/evaluate_clips.py', '--cfg', str(tmp_path / 'c.yaml'), '--dataset', fx['dataset'],
           '--db-dir', str(tmp_path), '--base-data', str(tmp_path / 'base'), '--vibe-ckpt', str(tmp_path / 'vibe.pth.tar'),
           '--vibe-layers', str(fx['L']), '--vibe-hidden', str(fx['H']), '--seqlen', str(fx['T'])]
    if fx['avg_filter']:
        cmd += ['--filter']                                          # evaluate.py --filter (lines 273-291)
    if world > 1:
        cmd += ['--gpus', str(world), '--backend', 'gloo', '--share-device0']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert line['data'] == 'real' and line['tables'] == 'files' and line['n_gpus'] == world and line['avg_filter'] == fx['avg_filter']
    if fx['dataset'] == '3dpw':
        assert line['vs_published']['mpjpe']['published'] == 84.6      # the published row of this config / set rides along
    assert set(line['metrics_mm']) == set(fx['final'])
    for k, v in fx['final'].items():
        assert abs(line['metrics_mm'][k] - v) < 1e-2, (k, line['metrics_mm'][k], v)
