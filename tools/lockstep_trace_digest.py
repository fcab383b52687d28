"""Digest of `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/driver_trace.py`: kernel time per lock-step of run_clips and the kernel
sequence of two individual steps.   python tools/lockstep_trace_digest.py <dir> > profiles/rNN_lockstep_trace.txt"""
import collections
import csv
import glob
import re
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = n.replace('tepose::', '')
    m = re.match(r'_ZN6tepose\d+([a-z_0-9]+?)I', n)
    return m.group(1) if m else n.split('(')[0].replace('void ', '')[:46]


ends = [i for i, r in enumerate(rows) if 'smpl_joints' in r['Kernel_Name']]       # one per forward; the first 8 belong to the warm-up pass
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[ends[8] + 1:]:
    k = short(r['Kernel_Name'])
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
steps = len(ends) - 9
tot = sum(v[1] for v in agg.values())
print('# rocprofv3 --kernel-trace of tools/driver_trace.py: run_clips over 37 clips of 300-1800 frames, T = 6, projection cache; %d lock-steps' % steps)
print('%-46s %8s %12s %10s' % ('kernel', 'calls', 'us per step', 'share'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print('%-46s %8d %12.2f %9.1f%%' % (k, v[0], v[1] / steps, 100 * v[1] / tot))
print('sum %.1f us per lock-step, %.2f launches per step' % (tot / steps, sum(v[0] for v in agg.values()) / steps))
for which in (40, 900):
    a, b = ends[which - 1] + 1, ends[which] + 1
    t0 = int(rows[a]['Start_Timestamp'])
    print('--- one step (#%d), %.1f us:' % (which, (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3))
    for r in rows[a:b]:
        print('  %-44s grid %7s x %s x %s  start %7.1f dur %6.1f' % (short(r['Kernel_Name']), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'],
                                                                    (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
