"""Phase timeline of wave 0 of sampled workgroups of the fused GRU step (gru_h3s16_kernel<0,2>; diagnostic build):
    tools/build_abl.sh gemm_h3s16 TEPOSE_S16_STAMPS s16stamps 1 ; TEPOSE_AMD_LIB=build/abl/lib_s16stamps1.so python tools/g16_stamps.py
Stamps (s_memtime): 0 entry, 1 first requests issued, 2 K loop done, per row tile of the cell update 3+3i loads issued / 4+3i loads landed /
5+3i math done + stores issued, 15 stores drained.  The LAST step launch of a forward is what remains in the buffer."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

dev = torch.device('cuda', 0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=synth.synthetic_smpl(0))
x = synthetic_windows_device(8192, 16, 3, dev)
with torch.no_grad():
    for _ in range(3):
        model.encoder(x)
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ['TEPOSE_AMD_LIB'])
buf = (ctypes.c_ulonglong * 256)()
raw.tepose_debug_g16_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.tepose_debug_g16_stamps(buf, 256) == 0
names = ['prologue requests', 'K loop'] + sum([['tile %d: issue loads' % i, 'tile %d: loads land' % i, 'tile %d: math + stores issued' % i] for i in range(4)], []) + ['stores drain']
idx = [1, 2] + sum([[3 + 3 * i, 4 + 3 * i, 5 + 3 * i] for i in range(4)], []) + [15]
t00 = min(buf[s * 16] for s in range(16) if buf[s * 16])
for s in range(16):
    st = [buf[s * 16 + k] for k in range(16)]
    if not st[0]:
        continue
    prev = st[0]
    parts = []
    for n, k in zip(names, idx):
        parts.append('%s %d' % (n.split(': ')[-1] if n.startswith('tile') else n, st[k] - prev))
        prev = st[k]
    print('wg %4d start +%7d | total %6d | ' % (s * 64 + 5, st[0] - t00, st[15] - st[0]) + ' | '.join(parts))
