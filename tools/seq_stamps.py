"""Per-step timeline of one workgroup of the persistent recurrent kernel (diagnostic build -DTEPOSE_SEQ_STAMPS,
library path in TEPOSE_AMD_LIB): wall-clock (100 MHz) stamps at 0 step start, 1 poll matched, 2 barrier left,
3 MFMA + LDS partials done (drained), 4 barrier left, 5 cell update + stores issued, 6 stores drained, 7 arrived."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(3 * 36 * 8, dtype=torch.int64, device='cuda')
os.environ['TEPOSE_SEQ_STAMP_PTR'] = str(buf.data_ptr())
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
smpl_np = synth.synthetic_smpl(0)
model, _, _ = build_model(2, 1024, seed=0, device='cuda', smpl_np=smpl_np)
x = synthetic_windows_device(B, T, 7, torch.device('cuda'))
with torch.no_grad():
    for _ in range(5):
        model.encoder(x)
torch.cuda.synchronize()
s = buf.cpu().view(3, 36, 8).double() / 100.0          # us ; the last launch (top layer) overwrote the first
names = ['start', 'poll', 'bar1', 'mfma', 'bar2', 'cell', 'drain', 'arrive']
for d in range(2):
    print('direction', d)
    r0 = s[d, 0]
    print('  step  0 (first step: element-wise, W_hh slices loading)  ' + '  '.join('%s %5.2f' % (names[k], float(r0[k] - r0[0])) for k in range(1, 8)) + '   | next start +%.2f' % float(s[d, 1, 0] - r0[0]))
    for st in range(1, T - 1):
        r = s[d, st]
        print('  step %2d  ' % st + '  '.join('%s %5.2f' % (names[k], float(r[k] - r[0])) for k in range(1, 8)) +
              '   | next start +%.2f' % float(s[d, st + 1, 0] - r[0]))
