"""Diagnostic (known issues: two processes sharing one GPU): tepose_smpl_fwd into a NaN-prefilled output, repeated; are wrong entries NaN (store never landed) or values?"""
import sys
import torch
sys.path.insert(0, sys.argv[1])
from tepose_amd import _lib, synth
from tepose_amd.testing import build_model
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(1, 64, seed=0, device='cuda', smpl_np=smpl_np, seqlen=5)
eng = model._engine
x4 = torch.from_numpy(synth.synthetic_windows(4, 5, 3)).cuda()
with torch.no_grad():
    model(x4)                      # packs everything
N = 20
pose = torch.from_numpy(synth.normal('probe_pose', (N, 72), std=0.3)).cuda()
betas = torch.from_numpy(synth.normal('probe_betas', (N, 10), std=0.5)).cuda()
ws = eng.workspace(10, 1, pose.device)
st = torch.cuda.current_stream().cuda_stream
bufs = [torch.empty(N, 6890, 3, device='cuda') for _ in range(3)]
jb = torch.empty(N, 49, 3, device='cuda')
def run(i):
    v = bufs[i % 3]
    v.fill_(float('nan'))
    _lib.check(eng.lib.tepose_smpl_fwd(eng.handle, 1, pose.data_ptr(), betas.data_ptr(), N, v.data_ptr(), jb.data_ptr(), ws.data_ptr(), ws.numel(), st), 'smpl')
    return v
ref = run(0).clone()
bad = nanbad = 0
for it in range(int(sys.argv[2])):
    v = run(it)
    if not torch.equal(v, ref):
        bad += 1
        ne = (v != ref)
        nn = int(torch.isnan(v).sum())
        nanbad += nn > 0
        if bad <= 8:
            idx = ne.reshape(N, -1).nonzero()
            print('iter %d: %d wrong, %d of them NaN (never written); person %s first idx %d' % (it, int(ne.sum()), nn, sorted(set(idx[:, 0].tolist())), int(idx[0, 1])), flush=True)
print('mismatches', bad, 'with NaN', nanbad)
