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


def traffic(fetch_path, write_path, proj_symbol, gru_symbol, out_json):
    """profiles/rNN_traffic_split.json from the FETCH_SIZE and WRITE_SIZE passes of tools/profile.sh: HBM-side bytes per launch of
    the dominant kernel (layer-0 projection, grid 131072 = 256 workgroups x 512) and of the fused 3-direction GRU step, FETCH
    x2-corrected (MI355X_MICROARCH.md, HBM section: the counter is in 32-byte units that report 64-byte requests once on gfx950),
    WRITE_SIZE as reported, both in KB.  The kernel SYMBOLS go into the file: bench.py refuses it as stale when the running
    library launches other kernels (tepose_kernel_info)."""
    import json

    def avg(path, counter, sym, grid=None):
        vals = defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == counter and sym.replace(' ', '') in r['Kernel_Name'].replace('tepose::', '').replace(' ', ''):
                vals[r['Grid_Size']].append(float(r['Counter_Value']))
        if grid is None:                      # the launch family with the largest grid (the 3-direction step)
            grid = max(vals, key=lambda g: int(g))
        v = vals[grid]
        v = v[1:] if len(v) > 2 else v        # drop the cold first launch
        return sum(v) / len(v), len(v), grid
    M, N, K = 131072, 9216, 2133
    pf, n1, g1 = avg(fetch_path, 'FETCH_SIZE', proj_symbol)
    pw, _, _ = avg(write_path, 'WRITE_SIZE', proj_symbol)
    gf, n2, g2 = avg(fetch_path, 'FETCH_SIZE', gru_symbol)
    gw, _, _ = avg(write_path, 'WRITE_SIZE', gru_symbol)
    alg = float(M) * K * 4 + float(N) * K * 4 + float(M) * N * 4
    B, Hp = 8192, 1024
    galg = 3 * (2.0 * B * Hp * 4 + 3.0 * Hp * Hp * 4 + 3.0 * B * Hp * 4)       # per direction: state in + out, W_hh once, gate pre-activations in
    d = {'kernel': '%s layer-0 input projection (M=%d,N=%d,K=%d), grid %s, %d steady-state launches' % (proj_symbol, M, N, K, g1, n1),
         'fetch_size_kb': pf, 'write_size_kb': pw, 'fetch_correction': 2.0,
         'traffic_bytes_per_launch': (2.0 * pf + pw) * 1024.0, 'algorithmic_bytes_per_launch': alg,
         'ratio': (2.0 * pf + pw) * 1024.0 / alg,
         'gru_step': {'kernel': '%s fused GRU step, 3 directions, B=%d, grid %s, %d launches' % (gru_symbol, B, g2, n2),
                      'fetch_size_kb': gf, 'write_size_kb': gw, 'traffic_bytes_per_launch': (2.0 * gf + gw) * 1024.0,
                      'algorithmic_bytes_per_launch': galg, 'ratio': (2.0 * gf + gw) * 1024.0 / galg,
                      'note': 'algorithmic = per direction: previous state in + new state out (fp32-equivalent 4 B/element each), W_hh once, gate pre-activations in'},
         'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of bench.py --steps 2 --warmup 1 (tools/profile.sh); digest: profiles/summarize.py traffic'}
    json.dump(d, open(out_json, 'w'), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == '__main__':
    {'trace': trace, 'pmc': pmc, 'sq': sq, 'pmcavg': pmcavg, 'traffic': traffic}[sys.argv[1]](*sys.argv[2:])
