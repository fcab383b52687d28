"""CPU: the kernel selection of the C library (csrc/api.hip select_kernels, exposed as tepose_select_kernels) pinned at every batch-class boundary.
Every threshold is a parity boundary -- another kernel family, another association of the K sums, another operand layout -- that used to be guarded by
the GPU suite only.  No device is needed: the function is pure host code of (knobs, L, hidden, B, T); the persistent small-batch
kernels plan with TEPOSE_ASSUME_CUS here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _select(L, H, cases, env=None):
    """{(B, T): dict} from a fresh process (knobs and the CU count are latched per process / per handle)."""
    code = ("import json, sys\nsys.path.insert(0, %r)\nfrom tepose_amd.engine import Engine\ne = Engine(%d, %d)\n"
            "print(json.dumps([[b, t, e.select_kernels(b, t)] for b, t in %r]))" % (ROOT, L, H, list(cases)))
    e = dict(os.environ, TEPOSE_ASSUME_CUS='256')
    for k in ('TEPOSE_EXACT_FP32', 'TEPOSE_LARGE_BATCH_KERNELS', 'TEPOSE_GRU_STATE', 'TEPOSE_S_MIN_B', 'TEPOSE_GI_BLK', 'TEPOSE_PERSISTENT'):
        e.pop(k, None)
    e.update(env or {})
    import json
    out = subprocess.run([sys.executable, '-c', code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return {(b, t): d for b, t, d in json.loads(out.stdout.strip().splitlines()[-1])}


@pytest.fixture(scope='module', autouse=True)
def _built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()


def test_class_boundaries_of_the_published_architecture():
    T = 16
    s = _select(2, 1024, [(1, T), (4, T), (5, T), (8, T), (9, T), (32, T), (64, T), (65, T), (128, T), (129, T), (511, T), (512, T), (639, T), (640, T),
                          (648, T), (656, T), (768, T), (8192, T), (8192, 6), (1360, 6), (1376, 6), (37, 6), (32, 6)])
    # input split: one workgroup per row up to 1024 rows (B * T), the 8-rows-per-block kernel above
    assert s[(4, T)]['input'] == 'split_rows_few_kernel' and s[(64, T)]['input'] == 'split_rows_few_kernel' and s[(65, T)]['input'] == 'split_rows_kernel'
    assert s[(37, 6)]['input'] == 'split_rows_few_kernel' and s[(8192, T)]['input'] == 'split_rows_kernel'
    # layer >= 1 projections: width-first kernel up to 192 real rows (three 64-row passes), tiles above
    assert s[(5, T)]['projection_l1'] == 'skinny_gemm_h3_kernel' and s[(8, T)]['projection_l1'] == 'skinny_gemm_h3_kernel'       # 80 / 128 rows
    assert s[(32, 6)]['projection_l1'] == 'skinny_gemm_h3_kernel' and s[(37, 6)]['projection_l1'] == 'gemm_h3_kernel'              # 192 / 222 rows
    assert s[(32, T)]['projection_l1'] == 'gemm_h3_kernel'
    # layer-0 projection: width-first kernel up to 128 rows | 128 x 128 tiles | 128 x 288 tiles where they need fewer whole rounds (from 512 rows) |
    # barrier-free persistent kernel from B * T = 8192
    assert s[(8, T)]['projection'] == 'skinny_gemm_h3_kernel' and s[(9, T)]['projection'] == 'gemm_h3_kernel'
    assert s[(32, T)]['projection'] == 'gemm_h3s_kernel<1, 3, 4, 3, 4>' and s[(64, T)]['projection'] == 'gemm_h3s_kernel<1, 3, 4, 3, 4>'
    assert s[(65, T)]['projection'] == 'gemm_h3_kernel'                   # 9 row tiles: 2 rounds of 288-wide tiles (~4.2) against 3 of 128-wide
    assert s[(511, T)]['projection'] == 'gemm_h3s_kernel<1, 3, 4, 3, 4>' and s[(512, T)]['projection'] == 'gemm_h3s_persist16c_kernel<0>'   # B * T = 8176 | 8192
    assert s[(1360, 6)]['projection'] != 'gemm_h3s_persist16c_kernel<0>' and s[(1376, 6)]['projection'] == 'gemm_h3s_persist16c_kernel<0>'  # 8160 | 8256 rows
    # recurrent part: persistent kernel up to 64 windows (granules up to 4) | width-first step up to 128 | 128-row tiles | scaled planes from 640
    assert s[(4, T)]['gru_step'] == 'gru_seq_kernel(granules)' and s[(5, T)]['gru_step'] == 'gru_seq_kernel' and s[(64, T)]['gru_step'] == 'gru_seq_kernel'
    assert s[(37, 6)]['gru_step'] == 'gru_seq_kernel'
    assert s[(65, T)]['gru_step'] == 'skinny_gru_h3_kernel' and s[(128, T)]['gru_step'] == 'skinny_gru_h3_kernel' and s[(129, T)]['gru_step'] == 'gemm_h3_kernel<GRU>'
    assert s[(639, T)]['gru_step'] == 'gemm_h3_kernel<GRU>' and s[(639, T)]['gi1_layout'] == 'row_major' and s[(639, T)]['projection_l1'] == 'gemm_h3_kernel'
    assert s[(640, T)]['gru_step'] == 'gru_step16_kernel<true>' and s[(640, T)]['gi1_layout'] == 'blocked' and s[(640, T)]['gru_step_l1'] == 'gru_step16_kernel<true>'
    assert s[(640, T)]['projection_l1'] == 'gemm_h3s_persist16c_kernel<1>' and s[(640, T)]['gru_first'] == 'gru_first16_kernel'
    # B % 128: the plane-fed step kernel needs full row tiles; B % 16: frame-major blocked layer-0 gate pre-activations need whole row tiles per frame
    assert s[(648, T)]['gru_step'] == 'gru_step16_kernel<false>' and s[(648, T)]['gi0_layout'] == 'row_major' and s[(648, T)]['gi1_layout'] == 'blocked'
    assert s[(656, T)]['gru_step'] == 'gru_step16_kernel<false>' and s[(656, T)]['gi0_layout'] == 'frame_major_blocked'
    assert s[(768, T)]['gru_step'] == 'gru_step16_kernel<true>' and s[(768, T)]['gi0_layout'] == 'frame_major_blocked'
    assert s[(640, T)]['gi0_layout'] == 'frame_major_blocked'
    # blend shapes: barrier-free persistent kernel from 512 persons; 1 - 4 persons: the one-launch SMPL kernel
    assert 'gemm_h3s_persist16c_kernel<1>' in s[(512, T)]['smpl'] and 'gemm_h3s_persist16c' not in s[(511, T)]['smpl']
    assert s[(4, T)]['smpl'] == 'smpl_small_kernel' and s[(5, T)]['smpl'] != 'smpl_small_kernel'
    # layer 0 keeps the fp32-state step where its gate pre-activations are row-major (B * T < 8192), layers >= 1 are blocked from 640 windows on
    s2 = _select(2, 1024, [(640, 4), (768, 8), (1024, 8)])
    assert s2[(640, 4)]['gru_step'] == 'gru_step16_kernel<false>' and s2[(640, 4)]['gru_step_l1'] == 'gru_step16_kernel<true>' and s2[(640, 4)]['gi0_layout'] == 'row_major'
    assert s2[(768, 8)]['gru_step'] == 'gru_step16_kernel<false>' and s2[(1024, 8)]['gru_step'] == 'gru_step16_kernel<true>'
    # cfg-C and the published window
    for key in ((8192, T), (8192, 6)):
        assert s[key]['projection'] == 'gemm_h3s_persist16c_kernel<0>' and s[key]['gru_step'] == 'gru_step16_kernel<true>'


def test_hidden_size_boundaries():
    # Hp % 256 / Hp <= 1024: the persistent recurrent kernel; Hp % 128: the 16-row first-step kernel; Hp % 288-wide tiles need (9 Hp) % 288 == 0 (always, Hp % 64 == 0)
    s = _select(2, 1000, [(8, 16), (2048, 4)])                 # Hp = 1024
    assert s[(8, 16)]['gru_step'] == 'gru_seq_kernel' and s[(2048, 4)]['gru_first'] == 'gru_first16_kernel'
    s = _select(2, 192, [(8, 16), (2048, 4), (2048, 8)])       # Hp = 192: not a multiple of 256 / 128
    assert s[(8, 16)]['gru_step'] == 'skinny_gru_h3_kernel' and s[(2048, 4)]['gru_first'] == 'gru_first_kernel'
    assert s[(2048, 4)]['gru_step'] == 'gru_step16_kernel<true>' and s[(2048, 8)]['gi0_layout'] == 'frame_major_blocked'
    s = _select(2, 2048, [(8, 16)])                            # Hp = 2048 > 1024: W_hh no longer fits the registers of one workgroup per 16 units
    assert s[(8, 16)]['gru_step'] == 'skinny_gru_h3_kernel'
    s = _select(1, 1024, [(64, 16), (8192, 16)])               # one layer: 2-direction launches, two-accumulator layer-0 projection at every size
    assert s[(64, 16)]['gru_step'] == 'gru_seq_kernel' and s[(8192, 16)]['projection'] == 'gemm_h3_kernel' and 'projection_l1' not in s[(64, 16)]


def test_named_knobs():
    c = [(64, 16), (640, 16), (8192, 16)]
    s = _select(2, 1024, c, {'TEPOSE_GRU_STATE': 'fp32'})
    assert s[(8192, 16)]['gru_step'] == 'gru_step16_kernel<false>' and s[(8192, 16)]['gi1_layout'] == 'blocked'
    s = _select(2, 1024, c, {'TEPOSE_LARGE_BATCH_KERNELS': 'twoacc'})
    assert s[(8192, 16)]['projection'] == 'gemm_h3_kernel' and s[(8192, 16)]['gru_step'] == 'gemm_h3_kernel<GRU>' and s[(8192, 16)]['gi0_layout'] == 'row_major'
    assert 'gemm_h3s' not in s[(8192, 16)]['smpl'] and s[(64, 16)]['projection'] == 'gemm_h3_kernel'
    s = _select(2, 1024, c, {'TEPOSE_EXACT_FP32': '1'})
    assert s[(8192, 16)]['projection'] == 'gemm_f32_kernel' and s[(8192, 16)]['gru_step'] == 'gru_step_kernel' and s[(64, 16)]['gru_step'] == 'skinny_gru_kernel'
    s = _select(2, 1024, c, {'TEPOSE_PERSISTENT': '0'})
    assert s[(64, 16)]['gru_step'] == 'skinny_gru_h3_kernel' and s[(64, 16)]['gru_first'] == 'gru_first_kernel'
    s = _select(2, 1024, c, {'TEPOSE_S_MIN_B': '2048'})
    assert s[(640, 16)]['gru_step'] == 'gemm_h3_kernel<GRU>' and s[(8192, 16)]['gru_step'] == 'gru_step16_kernel<true>'
    s = _select(2, 1024, c, {'TEPOSE_GI_BLK': '0'})
    assert s[(8192, 16)]['gi1_layout'] == 'row_major' and s[(8192, 16)]['gru_step'] == 'gru_step16_kernel<false>'
    # a chip with fewer CUs than a 3-direction layer needs workgroups: no persistent recurrent kernel
    s = _select(2, 1024, c, {'TEPOSE_ASSUME_CUS': '128'})
    assert s[(64, 16)]['gru_step'] == 'skinny_gru_h3_kernel'


def test_kernel_info_is_the_cfg_c_selection():
    code = ("import sys\nsys.path.insert(0, %r)\nfrom tepose_amd.engine import Engine\ne = Engine(2, 1024)\n"
            "k, s = e.kernel_info(), e.select_kernels(8192, 16)\nassert k == {'projection': s['projection'], 'gru_step': s['gru_step']}, (k, s)\nprint('ok')" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, TEPOSE_ASSUME_CUS='256'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-1500:]


def test_options_are_per_handle_not_per_process():
    """SURVEY 8b "no global mutable state" (VERDICT r5 item 7): two handles of ONE process with different thresholds select different kernels; a handle
    created later does not see another handle's options; the environment is read at tepose_create only."""
    code = r"""
import json, os, sys
sys.path.insert(0, %r)
from tepose_amd.engine import Engine
from tepose_amd import _lib
a, b = Engine(2, 1024), Engine(2, 1024)
assert a.get_option('S_MIN_B') == 640 and a.get_option('TEPOSE_SKINNY_H3_MAX_M') == 128 and a.get_option('NO_SUCH') == -1
b.set_option('S_MIN_B', 2048)
b.set_option('TEPOSE_SKINNY_H3_MAX_M', 64)
b.set_option('SEQ_MAX_M', 16)
os.environ['TEPOSE_S_MIN_B'] = '4096'                 # after creation: nobody re-reads it ...
out = {'a': [a.select_kernels(640, 16)['gru_step'], a.select_kernels(100, 16)['gru_step'], a.select_kernels(32, 16)['gru_step']],
       'b': [b.select_kernels(640, 16)['gru_step'], b.select_kernels(100, 16)['gru_step'], b.select_kernels(32, 16)['gru_step']]}
c = Engine(2, 1024)                                   # ... except the next tepose_create
out['c'] = [c.select_kernels(640, 16)['gru_step'], c.get_option('S_MIN_B'), c.get_option('SEQ_MAX_M')]
out['a_after'] = [a.get_option('S_MIN_B'), a.get_option('SEQ_MAX_M')]
rc = _lib.load().tepose_set_option(a.handle, b'NO_SUCH_OPTION', 1)
out['unknown_rc'] = rc
print(json.dumps(out))
""" % ROOT
    e = dict(os.environ, TEPOSE_ASSUME_CUS='256')
    for k in ('TEPOSE_S_MIN_B', 'TEPOSE_SKINNY_H3_MAX_M', 'TEPOSE_SEQ_MAX_M'):
        e.pop(k, None)
    import json
    out = subprocess.run([sys.executable, '-c', code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r['a'] == ['gru_step16_kernel<true>', 'skinny_gru_h3_kernel', 'gru_seq_kernel']
    assert r['b'] == ['gemm_h3_kernel<GRU>', 'gemm_h3_kernel<GRU>', 'skinny_gru_h3_kernel']       # S_MIN_B 2048 | SKINNY_H3_MAX_M 64 | SEQ_MAX_M 16
    assert r['c'] == ['gemm_h3_kernel<GRU>', 4096, 64] and r['a_after'] == [640, 64]
    assert r['unknown_rc'] == -1


def test_no_launcher_reads_the_environment():
    """No function-local static (or any other code outside tepose_create's options_from_env / read_env_knobs and the handle-less test entry points)
    reads a TEPOSE_* variable: grep of the sources."""
    import glob
    import re
    hits = []
    for f in sorted(glob.glob(os.path.join(ROOT, 'tepose_amd', 'csrc', '*'))):
        src = open(f).read()
        for mo in re.finditer(r'getenv\(', src):
            line = src.count('\n', 0, mo.start()) + 1
            hits.append((os.path.basename(f), line, src.splitlines()[line - 1].strip()))
    allowed = [h for h in hits if h[0] == 'gemm.hip' or (h[0] == 'api.hip' and ('read_env' in h[2] or 'TEPOSE_H3S' in h[2] or re.match(r'(const char\* )?e = getenv', h[2])))]
    assert hits == allowed, [h for h in hits if h not in allowed]
    assert not [h for h in hits if 'static' in h[2]], hits
