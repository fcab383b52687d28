"""Kernel trace target: ONE run_clips pass (37 clips of 300-1800 frames, T = 6, projection cache) after a warm-up pass on a short prefix.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_driver_trace -- python3 tools/driver_trace.py"""
import sys, os, time, torch
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
run_clips(model, [f[:12] for f in feats], inits, 6, J_regressor=J, keep=('kp_3d', 'verts'))
torch.cuda.synchronize(); t0 = time.perf_counter()
run_clips(model, feats, inits, 6, J_regressor=J, keep=('kp_3d', 'verts'))
torch.cuda.synchronize()
print('run_clips %.1f ms, %d lock-steps' % ((time.perf_counter() - t0) * 1e3, int(lens.max()) - 5))
