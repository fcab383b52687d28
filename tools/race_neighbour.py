"""Diagnostic for DESIGN.md section 10: a second PROCESS that runs ONE kind of work in a loop next to tools/race_probe_smpl.py, to find
out which neighbour the packed-fp32 anomaly needs:
   python tools/race_neighbour.py <repo> <seconds> <mode>
modes: smpl (tepose_smpl_fwd), fill (torch NaN fill of a [20,6890,3] buffer), equal (torch.equal of two such buffers: compare
kernel + sync + small D2H), gemm (tepose_gemm_h3_f32 20 x 20736 x 224), aa (tepose_rotmat_to_angle_axis), d2h (1.6 MB
device-to-host copies), idle (holds a context, launches nothing)"""
import sys
import time
import torch
sys.path.insert(0, sys.argv[1])
from tepose_amd import _lib, synth
from tepose_amd.testing import build_model
mode = sys.argv[3]
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(1, 64, seed=0, device='cuda', smpl_np=smpl_np, seqlen=5)
eng = model._engine
with torch.no_grad():
    model(torch.from_numpy(synth.synthetic_windows(4, 5, 3)).cuda())
N = 20
pose = torch.from_numpy(synth.normal('probe_pose', (N, 72), std=0.3)).cuda()
betas = torch.from_numpy(synth.normal('probe_betas', (N, 10), std=0.5)).cuda()
ws = eng.workspace(11, 1, pose.device)
st = torch.cuda.current_stream().cuda_stream
v = torch.empty(N, 6890, 3, device='cuda')
v2 = torch.zeros(N, 6890, 3, device='cuda')
jb = torch.empty(N, 49, 3, device='cuda')
gA = torch.randn(N, 224, device='cuda'); gW = torch.randn(20736, 224, device='cuda') * 0.01; gC = torch.empty(N, 20736, device='cuda')
gws = torch.empty(int(eng.lib.tepose_gemm_h3_workspace_bytes(N, 20736, 224)), dtype=torch.uint8, device='cuda')
rR = torch.randn(N * 24, 3, 3, device='cuda'); rA = torch.empty(N * 24, 3, device='cuda')
import ctypes
import os
hip = ctypes.CDLL('libamdhip64.so')
gws32 = torch.empty(int(eng.lib.tepose_gemm_workspace_bytes(20736, 224)), dtype=torch.uint8, device='cuda')
hA = torch.randn(512, 1024, device='cuda', dtype=torch.float16); hW = torch.randn(4096, 1024, device='cuda', dtype=torch.float16)
if mode == 'gemm_h3s':
    os.environ['TEPOSE_H3S'] = '1'
torch.cuda.synchronize()
t_end = time.time() + float(sys.argv[2])
n = 0
while time.time() < t_end:
    for _ in range(20):
        if mode == 'smpl':
            _lib.check(eng.lib.tepose_smpl_fwd(eng.handle, 1, pose.data_ptr(), betas.data_ptr(), N, v.data_ptr(), jb.data_ptr(), ws.data_ptr(), ws.numel(), st), 'smpl')
        elif mode == 'fill':
            v.fill_(float('nan'))
        elif mode == 'equal':
            torch.equal(v2, v2)
        elif mode == 'gemm':
            _lib.check(eng.lib.tepose_gemm_h3_f32(gA.data_ptr(), 224, gW.data_ptr(), 224, None, gC.data_ptr(), 20736, N, 20736, 224, gws.data_ptr(), gws.numel(), st), 'gemm')
        elif mode == 'aa':
            _lib.check(eng.lib.tepose_rotmat_to_angle_axis(rR.data_ptr(), N * 24, rA.data_ptr(), st), 'aa')
        elif mode == 'd2h':
            v2.cpu()
        elif mode == 'memset':
            assert hip.hipMemsetAsync(ctypes.c_void_p(gws.data_ptr()), 0, ctypes.c_size_t(gws.numel()), ctypes.c_void_p(st)) == 0
        elif mode == 'gemm_h3s':
            _lib.check(eng.lib.tepose_gemm_h3_f32(gA.data_ptr(), 224, gW.data_ptr(), 224, None, gC.data_ptr(), 20736, N, 20736, 224, gws.data_ptr(), gws.numel(), st), 'gemm')
        elif mode == 'gemm_f32':
            _lib.check(eng.lib.tepose_gemm_f32(gA.data_ptr(), 224, gW.data_ptr(), 224, None, gC.data_ptr(), 20736, N, 20736, 224, 0, gws32.data_ptr(), gws32.numel(), st), 'gemm32')
        elif mode == 'matmul':
            torch.matmul(gA, gW.t())
        elif mode == 'matmul16':
            torch.matmul(hA, hW.t())
        n += 1
    if mode == 'idle':
        time.sleep(0.05)
    else:
        torch.cuda.synchronize()
print('neighbour', mode, 'iterations', n)
