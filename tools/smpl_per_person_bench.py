"""DESIGN.md section 9 as a measurement: SMPL blend shapes + skinning as ONE WAVEFRONT PER PERSON (BASELINE.json's sketch)
against the shipped split (blend-shape GEMM on the matrix cores over 128-person tiles + skinning kernel), B = 64 and 8192."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd import _lib, synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(1, 64, seed=0, device=dev, smpl_np=smpl_np)
eng = model._engine
with torch.no_grad():
    model(torch.from_numpy(synth.synthetic_windows(2, 4, 1)).to(dev))
lib, st = eng.lib, torch.cuda.current_stream().cuda_stream
for N in (64, 8192):
    pose = (torch.randn(N, 72, device=dev) * 0.3).contiguous()
    betas = (torch.randn(N, 10, device=dev) * 0.5).contiguous()
    ws = eng.workspace((N + 1) // 2, 1, dev)
    va, vb = torch.empty(N, 6890, 3, device=dev), torch.empty(N, 6890, 3, device=dev)
    jb = torch.empty(N, 49, 3, device=dev)

    def shipped():
        _lib.check(lib.tepose_smpl_fwd(eng.handle, 1, pose.data_ptr(), betas.data_ptr(), N, va.data_ptr(), None, ws.data_ptr(),
                                       ws.numel(), st), 'smpl')

    def per_person():
        _lib.check(lib.tepose_smpl_fwd_per_person(eng.handle, pose.data_ptr(), betas.data_ptr(), N, vb.data_ptr(), ws.data_ptr(),
                                                  ws.numel(), st), 'per_person')

    res = {}
    for name, f, reps in (('shipped (prep + blend GEMM + skin)', shipped, 50), ('one wavefront per person', per_person, 5 if N > 1000 else 50)):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps * 1e3
    err = float((va - vb).abs().max())
    print('N=%5d  ' % N + '  |  '.join('%s %.3f ms' % kv for kv in res.items()) + '  |  max |diff| %.2e' % err, flush=True)
