"""Digest of tools/profile_small.sh's three passes for one shape into the text committed under profiles/:
    python tools/small_profile_digest.py <tag> <shape>        e.g.  r02c 1x16
per-kernel times from the kernel trace, then FETCH_SIZE / WRITE_SIZE per forward (sum over a forward's launches)."""
import csv
import glob
import re
import sys
from collections import defaultdict

sys.path.insert(0, 'profiles')
from summarize import short, trace  # noqa: E402

tag, shape = sys.argv[1], sys.argv[2]
log = open('gpurun_out/%s_%s_trace.log' % (tag, shape)).read()
m = re.findall(r'B=\s*\d+ T=\s*\d+\s+[\d.]+ ms/forward\s+[\d.]+ windows/s', log)
tr = glob.glob('gpurun_out/%s_%s_trace/*/*kernel_trace.csv' % (tag, shape))[0]
rows = list(csv.DictReader(open(tr)))
nf = max(sum(1 for r in rows if 'smpl_joints_kernel' in r['Kernel_Name']), 1)          # one per forward
print('# rocprofv3 --kernel-trace of `python3 tools/sweep.py %s` (%d forwards, shipped binary, round 3 final): %s'
      % (shape, nf, m[-1] if m else ''))
trace(tr)
for name, unit in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    cc = glob.glob('gpurun_out/%s_%s_%s/*/*counter_collection.csv' % (tag, shape, name))
    if not cc:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(cc[0])):
        if r['Counter_Name'] != unit:
            continue
        a = agg[short(r['Kernel_Name'])]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    nfw = max([v[0] for k, v in agg.items() if k.startswith('smpl_joints_kernel')] + [1])
    print('\n# %s per forward (KB, sum over the forward\'s launches; kernels with >= 1 launch per forward)' % unit)
    tot = 0.0
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if n < nfw or k.startswith(('at::', 'pack', 'csr', 'absmax', '__amd', 'dmm', 'd2f')):
            continue
        print('%-60s x%-3d %12.1f KB' % (k[:60], round(n / nfw), v / nfw))
        tot += v / nfw
    print('total %.1f MB per forward%s' % (tot / 1e3, ' (x2 for FETCH on gfx950)' if unit == 'FETCH_SIZE' else ''))
