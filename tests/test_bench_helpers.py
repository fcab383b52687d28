"""CPU: bench.py's bookkeeping that must not lie -- a committed PMC traffic profile is quoted only for the kernel it was taken on."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_profile_is_refused_when_it_names_another_kernel():
    files = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_traffic_split.json'))
    assert files
    d = json.load(open(os.path.join(ROOT, 'profiles', files[-1])))
    proj = d['kernel'].split(' ')[0]
    gru = d['gru_step']['kernel'].split(' fused')[0]
    tr, src = bench.pmc_traffic(8192, 16, True, {'projection': proj, 'gru_step': gru})
    assert tr == d['traffic_bytes_per_launch'] and 'constant from profiles/' in src and bench.pmc_traffic.gru is not None
    tr, src = bench.pmc_traffic(8192, 16, True, {'projection': 'some_other_kernel<0>', 'gru_step': gru})
    assert tr is None and src.startswith('STALE') and bench.pmc_traffic.gru is not None
    tr, src = bench.pmc_traffic(8192, 16, True, {'projection': proj, 'gru_step': 'another_step_kernel'})
    assert tr == d['traffic_bytes_per_launch'] and bench.pmc_traffic.gru is None
    tr, src = bench.pmc_traffic(4096, 16, True, {'projection': proj, 'gru_step': gru})
    assert tr is None                                          # another batch than the profiled one


def test_newest_traffic_profile_matches_the_default_kernels():
    """The committed profile must describe what a default build launches today (tepose_kernel_info of a fresh handle)."""
    import __graft_entry__ as g
    g.build()
    from tepose_amd.engine import Engine
    ki = Engine(2, 1024).kernel_info()
    tr, src = bench.pmc_traffic(8192, 16, True, ki)
    assert tr is not None, src
    assert bench.pmc_traffic.gru is not None


def test_physical_core_count_is_sane():
    n = bench.physical_cores()
    assert n is None or 1 <= n <= (os.cpu_count() or 1)
