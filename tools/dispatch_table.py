"""Print the batch class -> kernels table of DESIGN.md section 0 from the library's own selection function (tepose_select_kernels; no GPU needed):
    TEPOSE_ASSUME_CUS=256 python tools/dispatch_table.py [n_layers hidden T]"""
import os
import sys

os.environ.setdefault('TEPOSE_ASSUME_CUS', '256')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tepose_amd.engine import Engine  # noqa: E402

L, H, T = (int(v) for v in (sys.argv[1:4] + ['2', '1024', '16'][len(sys.argv) - 1:]))
e = Engine(L, H)
rows, last = [], None
for B in list(range(1, 1301)) + [2048, 4096, 8192]:
    d = e.select_kernels(B, T)
    key = tuple(sorted(d.items()))
    if key != last:
        rows.append([B, B, d])
        last = key
    else:
        rows[-1][1] = B
cols = ['input', 'projection', 'gi0_layout', 'gru_step', 'gru_first', 'projection_l1', 'gi1_layout', 'gru_step_l1', 'smpl']
print('| B (T = %d, L = %d, H = %d) | ' % (T, L, H) + ' | '.join(cols) + ' |')
print('|' + '---|' * (len(cols) + 1))
for lo, hi, d in rows:
    print('| %s | ' % (str(lo) if lo == hi else '%d – %d' % (lo, hi)) + ' | '.join(d.get(c, '–').replace('_kernel', '') for c in cols) + ' |')
