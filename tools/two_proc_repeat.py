"""Two PROCESSES on one GPU, each repeating the same large-batch encoder forward and comparing every result bit for bit with its first one
(the way the packed-fp32 erratum of DESIGN.md section 10 was found: wrong results that only appear next to another process's LDS + MFMA
kernels).  Parent: python tools/two_proc_repeat.py [seconds] [B]   (spawns the two children before touching the GPU itself)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    sys.path.insert(0, ROOT)
    from bench import synthetic_windows_device
    from tepose_amd import synth
    from tepose_amd.testing import build_model
    secs, B, seed = float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device('cuda', 0)
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(2, 1024, seed=seed, device=dev, smpl_np=smpl_np)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    x = synthetic_windows_device(B, 16, 3 + seed, dev)
    with torch.no_grad():
        ref = model(x, J_regressor=J)[0]
        ref = {k: v.clone() for k, v in ref.items()}
    torch.cuda.synchronize()
    n = bad = 0
    t_end = time.time() + secs
    with torch.no_grad():
        while time.time() < t_end:
            out = model(x, J_regressor=J)[0]
            torch.cuda.synchronize()
            n += 1
            for k in ('theta', 'verts', 'kp_3d'):
                if not torch.equal(out[k], ref[k]):
                    bad += 1
                    print('MISMATCH forward %d output %s max|diff| %.3e' % (n, k, float((out[k] - ref[k]).abs().max())), flush=True)
    print('child seed %d: %d forwards of B=%d, %d mismatches, finite %s' % (seed, n, B, bad, bool(torch.isfinite(ref['verts']).all())), flush=True)
    sys.exit(1 if bad else 0)
secs = sys.argv[1] if len(sys.argv) > 1 else '30'
B = sys.argv[2] if len(sys.argv) > 2 else '2048'
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'child', secs, B, str(i)]) for i in range(2)]
rc = [p.wait() for p in procs]
print('two_proc_repeat:', 'OK' if not any(rc) else 'FAILED', rc)
sys.exit(1 if any(rc) else 0)
