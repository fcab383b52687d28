"""Diagnostic for DESIGN.md section 10, library built WITH packed fp32 ops and -DTEPOSE_SKIN_DIAG=2: the skinning kernel re-derives the first
row of every vertex's blended transform with scalar FMAs from a second LDS read and records the lanes whose PACKED accumulators
differ -- operands included, so the host can tell a wrong operand from a wrong result:
   TEPOSE_AMD_LIB=build/abl/lib_pkdiag2.so python tools/race_probe_smpl_check.py <repo> <iterations>     (run two at once)"""
import ctypes
import os
import struct
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
N = 20
pose = torch.from_numpy(synth.normal('probe_pose', (N, 72), std=0.3)).cuda()
betas = torch.from_numpy(synth.normal('probe_betas', (N, 10), std=0.5)).cuda()
ws = eng.workspace(11, 1, pose.device)
st = torch.cuda.current_stream().cuda_stream
v = torch.empty(N, 6890, 3, device='cuda')
jb = torch.empty(N, 49, 3, device='cuda')
raw = ctypes.CDLL(os.environ['TEPOSE_AMD_LIB'])
raw.tepose_debug_skin_check.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
n = 16 + 64 * 32
buf = (ctypes.c_uint * n)()
f = lambda u: struct.unpack('f', struct.pack('I', int(u)))[0]   # noqa: E731


def run():
    v.fill_(float('nan'))
    _lib.check(eng.lib.tepose_smpl_fwd(eng.handle, 1, pose.data_ptr(), betas.data_ptr(), N, v.data_ptr(), jb.data_ptr(), ws.data_ptr(), ws.numel(), st), 'smpl')
    torch.cuda.synchronize()
    return v.cpu().numpy()


ref = run()
raw.tepose_debug_skin_check(buf, n, 1)
bad_calls = flagged_calls = shown = 0
for it in range(int(sys.argv[2])):
    got = run()
    wrong = bool((got != ref).any())
    raw.tepose_debug_skin_check(buf, n, 1)
    cnt = int(buf[0])
    bad_calls += wrong
    flagged_calls += cnt > 0
    if (wrong or cnt) and shown < 4:
        shown += 1
        print('iter %d: output wrong %s, lanes flagged by the in-kernel check %d' % (it, wrong, cnt))
        for s in range(min(cnt, 3)):
            d = buf[16 + s * 32: 16 + s * 32 + 32]
            t = [f(d[2 + e]) for e in range(4)]
            u = [f(d[6 + e]) for e in range(4)]
            w = [f(d[10 + k]) for k in range(4)]
            q0 = [f(d[18 + k]) for k in range(4)]
            print('   person %d vertex %d (lane %d): packed row %s | scalar row %s | weights %s | joint offsets %s | first element of each joint row as re-read %s'
                  % (d[0], d[1], d[1] % 64, np.round(t, 5), np.round(u, 5), np.round(w, 4), list(d[14:18]), np.round(q0, 5)))
            print('      scalar row[0] recomputed on the host from the recorded operands: %.6f' % sum(np.float32(w[k]) * np.float32(q0[k]) for k in range(4)))
print('calls %s: wrong outputs %d, calls with flagged lanes %d' % (sys.argv[2], bad_calls, flagged_calls))
