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


def sq(counter_path, trace_path):
    """SQ pass (tools/profile.sh): clock from GRBM_GUI_ACTIVE (summed over the 8 XCDs), MFMA busy share of
    the 1024 SIMDs, wait shares of the wave cycles; durations from the same run's kernel trace."""
    dur = {}
    for r in csv.DictReader(open(trace_path)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    per = defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(counter_path)):
        per[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        meta[r['Dispatch_Id']] = (short(r['Kernel_Name']), r['Grid_Size'])
    agg = defaultdict(list)
    for d, c in per.items():
        if d in dur and 'GRBM_GUI_ACTIVE' in c:
            agg[meta[d]].append((dur[d], c))
    print('%-44s %9s %4s %10s %8s %9s %9s %9s' % ('kernel', 'grid', 'n', 'dur_us', 'clk_GHz', 'mfma_busy',
                                                    'wait_any', 'wait_inst'))
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        if not k.startswith(('gemm', 'gru')):
            continue
        n = len(v)
        du = sum(x[0] for x in v) / n
        f = lambda name: sum(x[1].get(name, 0.) for x in v) / n
        cyc = f('GRBM_GUI_ACTIVE') / 8
        print('%-44s %9s %4d %10.1f %8.3f %9.3f %9.3f %9.3f' % (
            k, g, n, du, cyc / du / 1e3, f('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc),
            f('SQ_WAIT_ANY') / max(f('SQ_WAVE_CYCLES'), 1), f('SQ_WAIT_INST_ANY') / max(f('SQ_WAVE_CYCLES'), 1)))
    print('# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles at the measured clock); '
          'wait_* = share of SQ_WAVE_CYCLES')


def pmcavg(counter_path, trace_path):
    """Any counter set: per kernel (recurrent / GEMM kernels only) the average duration and the average of every counter,
    plus each counter per microsecond."""
    dur = {}
    for r in csv.DictReader(open(trace_path)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    per = defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(counter_path)):
        per[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        meta[r['Dispatch_Id']] = (short(r['Kernel_Name']), r['Grid_Size'])
    agg = defaultdict(list)
    for d, c in per.items():
        if d in dur:
            agg[meta[d]].append((dur[d], c))
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        if not ('gru' in k or 'gemm_h3s' in k):
            continue
        n = len(v)
        du = sum(x[0] for x in v) / n
        names = sorted({c for x in v for c in x[1]})
        print('%-44s grid %9s n %4d dur_us %10.1f' % (k, g, n, du))
        for c in names:
            val = sum(x[1].get(c, 0.) for x in v) / n
            print('    %-36s %16.1f   per_us %12.1f' % (c, val, val / du))


if __name__ == '__main__':
    {'trace': trace, 'pmc': pmc, 'sq': sq, 'pmcavg': pmcavg}[sys.argv[1]](*sys.argv[2:])
