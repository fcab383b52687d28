"""Median duration per (kernel, grid) of a rocprofv3 kernel trace:  python tools/kernel_medians.py <kernel_trace.csv> [name substring ...]"""
import collections
import csv
import sys

agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]):
        continue
    agg[(n[:70], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print('%-72s grid %9s  n %4d  median %9.1f us  total %10.1f us' % (k[0], k[1], len(v), v[len(v) // 2], sum(v)))
