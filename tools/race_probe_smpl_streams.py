"""Diagnostic for DESIGN.md section 10: does the packed-fp32 anomaly need two PROCESSES (context switching) or only two kernels
co-resident on the chip?  ONE process, S streams, each running tepose_smpl_fwd (own handle, own workspace, own outputs) in a
loop; every result compared bitwise with the stream's first:
   python tools/race_probe_smpl_streams.py <repo> <iterations> [streams = 2] [N persons = 20] [neighbour]
neighbour (optional): what the LAST HALF of the streams runs instead of tepose_smpl_fwd (those streams are not checked):
   fill (a NaN fill of the output buffer), gemm (tepose_gemm_h3_f32, 20 x 20736 x 224: the blend-shape product's shape),
   aa (tepose_rotmat_to_angle_axis), sleep (nothing)"""
import sys
import torch
sys.path.insert(0, sys.argv[1])
from tepose_amd import _lib, synth
from tepose_amd.testing import build_model
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
smpl_np = synth.synthetic_smpl(0)
pose = torch.from_numpy(synth.normal('probe_pose', (N, 72), std=0.3)).cuda()
betas = torch.from_numpy(synth.normal('probe_betas', (N, 10), std=0.5)).cuda()
ctx = []
for s in range(S):
    model, _, _ = build_model(1, 64, seed=0, device='cuda', smpl_np=smpl_np, seqlen=5)
    with torch.no_grad():
        model(torch.from_numpy(synth.synthetic_windows(4, 5, 3)).cuda())
    eng = model._engine
    ws = torch.empty(int(eng.lib.tepose_workspace_bytes(eng.handle, max(10, N // 2 + 1), 1)), dtype=torch.uint8, device='cuda')
    ctx.append(dict(model=model, eng=eng, ws=ws, st=torch.cuda.Stream(), out=[torch.empty(N, 6890, 3, device='cuda') for _ in range(2)],
                    jb=torch.empty(N, 49, 3, device='cuda'), ref=None, bad=0, lanes=[0, 0, 0, 0]))
torch.cuda.synchronize()


NEIGH = sys.argv[5] if len(sys.argv) > 5 else ''
gA = torch.randn(N, 224, device='cuda')
gW = torch.randn(20736, 224, device='cuda') * 0.01
gC = [torch.empty(N, 20736, device='cuda') for _ in range(S)]
lib0 = ctx[0]['eng'].lib
gws = [torch.empty(int(lib0.tepose_gemm_h3_workspace_bytes(N, 20736, 224)), dtype=torch.uint8, device='cuda') for _ in range(S)]
rR = torch.randn(N * 24, 3, 3, device='cuda')
rA = [torch.empty(N * 24, 3, device='cuda') for _ in range(S)]


def launch(c, i):
    v = c['out'][i % 2]
    k = ctx.index(c)
    if NEIGH and k >= S // 2:
        with torch.cuda.stream(c['st']):
            for _ in range(3):
                if NEIGH == 'fill':
                    v.fill_(float('nan'))
                elif NEIGH == 'gemm':
                    _lib.check(lib0.tepose_gemm_h3_f32(gA.data_ptr(), 224, gW.data_ptr(), 224, None, gC[k].data_ptr(), 20736, N, 20736, 224,
                                                       gws[k].data_ptr(), gws[k].numel(), c['st'].cuda_stream), 'gemm')
                elif NEIGH == 'aa':
                    _lib.check(lib0.tepose_rotmat_to_angle_axis(rR.data_ptr(), N * 24, rA[k].data_ptr(), c['st'].cuda_stream), 'aa')
        return v if c['ref'] is None else c['ref']
    with torch.cuda.stream(c['st']):
        v.fill_(float('nan'))
        _lib.check(c['eng'].lib.tepose_smpl_fwd(c['eng'].handle, 1, pose.data_ptr(), betas.data_ptr(), N, v.data_ptr(), c['jb'].data_ptr(),
                                                c['ws'].data_ptr(), c['ws'].numel(), c['st'].cuda_stream), 'smpl')
    return v


for c in ctx:
    r = launch(c, 0)
    torch.cuda.synchronize()
    c['ref'] = r.clone()
torch.cuda.synchronize()
for it in range(int(sys.argv[2])):
    vs = [launch(c, it) for c in ctx]           # queued on every stream before anything is waited for
    for c, v in zip(ctx, vs):
        c['st'].synchronize()
        if not torch.equal(v, c['ref']):
            c['bad'] += 1
            ne = (v != c['ref']).nonzero()
            for q in range(4):
                c['lanes'][q] += int(((ne[:, 1] % 64) // 16 == q).sum())
print('neighbour %r, streams %d, N %d: mismatching calls per stream %s; wrong floats by quarter wave %s' % (NEIGH, S, N, [c['bad'] for c in ctx], [c['lanes'] for c in ctx]))
