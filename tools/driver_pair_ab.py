import sys, os, json, time, torch
sys.path.insert(0, ".")
from tepose_amd import synth
from tepose_amd.testing import build_model
from tepose_amd.driver import run_clips
dev = torch.device("cuda", 0)
smpl_np = synth.synthetic_smpl(0)
model, state, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np, seqlen=6)
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
lens = (300 + 1500 * synth.uniform01('evalclips', 37)).astype(int)
feats = [torch.from_numpy(synth.synthetic_windows(1, int(n), 100 + i)[0, :, :2048].copy()).to(dev) for i, n in enumerate(lens)]
inits = [torch.from_numpy(synth.synthetic_windows(1, 6, 200 + i)[0, :5, 2048:].copy()).to(dev) for i in range(37)]
steps = int(lens.max()) - 5
res = {}
for rnd in range(3):
    for name, env in (('step', '1'), ('pair', '2'), ('two', '0'), ('nocache', None)):
        os.environ['TEPOSE_DRIVER_PAIR'] = env or '1'
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run_clips(model, feats, inits, 6, J_regressor=J, keep=('kp_3d', 'verts'), cache_projections=(env is not None))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.setdefault(name, []).append(dt / steps * 1e3)
print({k: ['%.4f' % v for v in vs] for k, vs in res.items()}, 'ms per lock-step (37 clips, T = 6)')
# the same three loops after the benchmark batch went through the same handle (workspace grown to B = 8192, T = 16)
if len(sys.argv) > 1 and sys.argv[1] == 'after-big':
    xb = torch.from_numpy(synth.synthetic_windows(64, 16, 3)).to(dev).repeat(128, 1, 1)
    model(xb, J_regressor=J)
    torch.cuda.synchronize()
    del xb
    res = {}
    for rnd in range(2):
        for name, env in (('step', '1'), ('nocache', None)):
            os.environ['TEPOSE_DRIVER_PAIR'] = env or '1'
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run_clips(model, feats, inits, 6, J_regressor=J, keep=('kp_3d', 'verts'), cache_projections=(env is not None))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res.setdefault(name, []).append(dt / steps * 1e3)
    print('after a B = 8192, T = 16 forward on the same handle:', {k: ['%.4f' % v for v in vs] for k, vs in res.items()})
