"""Diagnostic for DESIGN.md section 10 (two processes sharing one GPU, packed-fp32 build): like race_probe_smpl.py, but for the first
mismatching calls it prints WHAT is wrong -- which lanes of which workgroup, which coordinate, wrong and right values, and whether
the wrong value is explained by a partial / stale computation:
   python tools/race_probe_smpl2.py <repo> <iterations> [N persons]"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from tepose_amd import _lib, synth
from tepose_amd.testing import build_model
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(1, 64, seed=0, device='cuda', smpl_np=smpl_np, seqlen=5)
eng = model._engine
with torch.no_grad():
    model(torch.from_numpy(synth.synthetic_windows(4, 5, 3)).cuda())
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pose = torch.from_numpy(synth.normal('probe_pose', (N, 72), std=0.3)).cuda()
betas = torch.from_numpy(synth.normal('probe_betas', (N, 10), std=0.5)).cuda()
ws = eng.workspace(max(10, N // 2 + 1), 1, pose.device)
st = torch.cuda.current_stream().cuda_stream
bufs = [torch.empty(N, 6890, 3, device='cuda') for _ in range(3)]
jb = torch.empty(N, 49, 3, device='cuda')


def run(i):
    v = bufs[i % 3]
    v.fill_(float('nan'))
    _lib.check(eng.lib.tepose_smpl_fwd(eng.handle, 1, pose.data_ptr(), betas.data_ptr(), N, v.data_ptr(), jb.data_ptr(), ws.data_ptr(), ws.numel(), st), 'smpl')
    return v


ref = run(0).clone()
refc = ref.cpu().numpy()
bad = shown = 0
lanes = {}
for it in range(int(sys.argv[2])):
    v = run(it)
    if not torch.equal(v, ref):
        bad += 1
        got = v.cpu().numpy()
        ne = np.argwhere(got != refc)
        for p, vert, c in ne:
            lanes[(int(vert) % 64) // 16] = lanes.get((int(vert) % 64) // 16, 0) + 1
        if shown < 5:
            shown += 1
            p, vert, c = ne[0]
            v0 = int(vert) // 16 * 16
            print('iter %d: %d wrong floats, coordinates %s, persons %s' % (it, len(ne), sorted(set(ne[:, 2].tolist())), sorted(set(ne[:, 0].tolist()))[:8]))
            print('  first group: person %d vertices %d..%d (workgroup %d, wave %d, lanes %d..%d)' % (p, v0, v0 + 15, v0 // 256, v0 % 256 // 64, v0 % 64, v0 % 64 + 15))
            print('  wrong x:', np.array2string(got[p, v0:v0 + 16, 0], precision=5, max_line_width=200))
            print('  right x:', np.array2string(refc[p, v0:v0 + 16, 0], precision=5, max_line_width=200))
            print('  right y:', np.array2string(refc[p, v0:v0 + 16, 1], precision=5, max_line_width=200))
            print('  wrong-right:', np.array2string(got[p, v0:v0 + 16, 0] - refc[p, v0:v0 + 16, 0], precision=5, max_line_width=200), flush=True)
print('mismatches', bad, 'wrong floats by quarter wave (lanes 0-15, 16-31, 32-47, 48-63):', [lanes.get(q, 0) for q in range(4)])
