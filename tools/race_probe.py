"""Diagnostic (known issues: two processes sharing one GPU): repeat each module of the small evaluation pipeline and compare bitwise with its first result (run it while
another GPU process is active to look for timing-dependent results).  argv[1] = package root, argv[2] = iterations"""
import sys

root, iters = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, root)
import torch  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.smpl import SMPL  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402
from tepose_amd.vibe import VIBE  # noqa: E402

T = 5
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(1, 64, seed=0, device='cuda', smpl_np=smpl_np, seqlen=T)
vstate = synth.synthetic_vibe_state_dict(1, 64, 1)
mean = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0], 'cam': vstate['regressor.init_cam'][0]}
vibe = VIBE(seqlen=T, n_layers=1, hidden_size=64, add_linear=True, use_residual=True, pretrained='', smpl=SMPL.from_tables(smpl_np),
            smpl_mean_params=mean)
sd = vibe.state_dict()
for k, v in vstate.items():
    sd[k] = torch.from_numpy(v)
vibe.load_state_dict(sd)
vibe = vibe.cuda().eval()
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
x4 = torch.from_numpy(synth.synthetic_windows(4, T, 3)).cuda()
x3 = torch.from_numpy(synth.synthetic_windows(3, T, 4)).cuda()
xv = x4[:, :, :2048].contiguous()
feat = torch.from_numpy(synth.normal('probe', (20, 2048), std=0.5)).cuda()
eng = model._engine
pose20 = torch.from_numpy(synth.normal('probe_pose', (20, 72), std=0.3)).cuda()
betas20 = torch.from_numpy(synth.normal('probe_betas', (20, 10), std=0.5)).cuda()


def cached():
    ring = torch.empty(4, T - 1, eng.gate_width, device='cuda')
    newest = torch.empty(4, eng.gate_width, device='cuda')
    pws = torch.empty(int(eng.lib.tepose_project_frames_workspace_bytes(eng.handle, 4)) if hasattr(eng.lib, 'tepose_project_frames_workspace_bytes') else 4 * 2144 * 4, dtype=torch.uint8, device='cuda')
    F, TH = x4[:, :, :2048].contiguous(), x4[:, :, 2048:].contiguous()
    for t in range(T - 1):
        eng.project_frames(F[:, t].data_ptr(), F.stride(0), TH[:, t].data_ptr(), TH.stride(0), 4, ring[:, t].data_ptr(), ring.stride(0), pws)
    eng.project_frames(F[:, T - 1].data_ptr(), F.stride(0), None, TH.stride(0), 4, newest.data_ptr(), newest.stride(0), pws)
    return eng.forward_cached(ring, 0, newest, 4, T, J)


mods = {
    'tepose_fwd_B4': lambda: model(x4, J_regressor=J)[0],
    'tepose_fwd_B3': lambda: model(x3, J_regressor=J)[0],
    'encoder_B4': lambda: {'f': model.encoder(x4)},
    'regressor_N20': lambda: model.regressor(feat, J_regressor=J)[0],
    'vibe_B4': lambda: vibe(xv, J_regressor=J)[-1],
    'cached_B4': cached,
    'smpl_only_N20': lambda: dict(zip(('verts', 'joints'), eng.smpl_fwd(pose20, betas20, True))),
}
with torch.no_grad():
    ref = {k: {n: t.clone() for n, t in f().items()} for k, f in mods.items()}
    bad = {k: 0 for k in mods}
    for it in range(iters):
        for k, f in mods.items():
            out = f()
            if any(not torch.equal(out[n], ref[k][n]) for n in out):
                bad[k] += 1
                d = {n: (float((out[n] - ref[k][n]).abs().max()), int((out[n] != ref[k][n]).sum())) for n in out if not torch.equal(out[n], ref[k][n])}
                print('iter %d: %s differs: %s' % (it, k, d), flush=True)
                if 'verts' in d and bad[k] <= 4:
                    ne = (out['verts'] != ref[k]['verts']).reshape(out['verts'].shape[0] if out['verts'].dim() == 3 else -1, 6890 * 3) if out['verts'].dim() == 3 else (out['verts'] != ref[k]['verts']).reshape(-1, 6890 * 3)
                    idx = ne.nonzero()
                    print('   persons', sorted(set(idx[:, 0].tolist())), 'flat float index in the person', idx[:, 1].tolist()[:40], flush=True)
print('mismatches', bad)
