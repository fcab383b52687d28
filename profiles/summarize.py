"""Digest rocprofv3 CSV output (kernel trace / PMC counter collection) into the short
per-kernel tables committed under profiles/.  Usage:
    python profiles/summarize.py trace  <kernel_trace.csv>
    python profiles/summarize.py pmc    <counter_collection.csv>
Only dispatches of the benchmark batch are kept (grid size filter: the tiny B=2 packing
forward and torch's own kernels are listed separately)."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace('tepose::', '')
    for cut in ('(', ):
        i = name.find(cut)
        if i > 0 and not name.startswith('void at'):
            name = name[:i]
    return name.replace('void ', '')[:70]


def trace(path):
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(list)
    for r in rows:
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        grid = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        agg[(short(r['Kernel_Name']), grid)].append(dur)
    tot = sum(sum(v) for v in agg.values())
    print('%-52s %10s %6s %12s %12s %7s' % ('kernel', 'grid', 'calls', 'avg_us', 'total_us', '%'))
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / tot < 0.0005:
            continue
        print('%-52s %10s %6d %12.1f %12.1f %7.2f' % (k, g, len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
    print('total_us %.1f' % tot)


def pmc(path):
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(list)
    for r in rows:
        agg[(short(r['Kernel_Name']), r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    print('%-52s %10s %-12s %6s %16s' % ('kernel', 'grid', 'counter', 'calls', 'avg_value'))
    for (k, g, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if not k.startswith(('gemm', 'gru', 'smpl', 'pad', 'init')):
            continue
        print('%-52s %10s %-12s %6d %16.1f' % (k, g, c, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    {'trace': trace, 'pmc': pmc}[sys.argv[1]](sys.argv[2])
