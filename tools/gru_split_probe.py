"""Probe for the "GEMM || cell update on two streams" idea: does an HBM-streaming element-wise kernel on a
second stream run BESIDE the step kernel's K loops, or do the two meet in the CU's vector-memory path as the fused update does?
Round 5 ran it with a timing-only build whose step kernel had no cell update (an ablation macro of gemm_h3s16.hip, a file that no longer exists: the record
is profiles/r05_split_probe.txt; HEAD has no such switch -- to repeat the probe, stub the update loop of csrc/gru_step16.hip in a private build and point
TEPOSE_AMD_LIB at it) and with the shipped library:
    python tools/gru_split_probe.py [B]
Prints, per configuration: the library's own hipEvent times of the layer-0 projection and of the recurrent part of one encoder
forward, alone and with a stream of torch.add(a, b, out=c) launches (3 x 100 MB each: the bytes a split-off update kernel of one
direction step would move, h W_hh^T round trip included) running beside it, and that stream's TB/s alone / beside."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device('cuda', 0)
model, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=synth.synthetic_smpl(0))
x = synthetic_windows_device(B, 16, 3, dev)
eng = model._engine
n_el = 25 * 1024 * 1024                       # 100 MB fp32 tensors
a, b, c = (torch.randn(n_el, device=dev) for _ in range(3))
s2 = torch.cuda.Stream()
bytes_per_add = 3 * n_el * 4


def enc(reps):
    eng.profile_enable(True)
    for _ in range(reps):
        model.encoder(x)


def read():
    k_ms, k_n, _ = eng.profile_read()
    g_ms, g_n, _ = eng.profile_read_gru()
    eng.profile_enable(False)
    return k_ms / max(k_n, 1), g_ms / max(g_n, 1)


with torch.no_grad():
    for _ in range(2):
        model.encoder(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); enc(4); torch.cuda.synchronize(); t_enc = (time.perf_counter() - t0) / 4 * 1e3
    p0, g0 = read()
    # element-wise stream alone
    with torch.cuda.stream(s2):
        for _ in range(10):
            torch.add(a, b, out=c)
    torch.cuda.synchronize()
    n_add = 400
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        for _ in range(n_add):
            torch.add(a, b, out=c)
    torch.cuda.synchronize()
    t_el = time.perf_counter() - t0
    bw_alone = n_add * bytes_per_add / t_el / 1e12
    # both: the element-wise stream long enough to cover 4 encoder forwards
    n_cov = int(4 * t_enc * 1e-3 * 2.0 / (bytes_per_add / (bw_alone * 1e12))) + 50
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        ev0.record(s2)
        for _ in range(n_cov):
            torch.add(a, b, out=c)
        ev1.record(s2)
    enc(4)
    cur = torch.cuda.current_stream()
    e_done = torch.cuda.Event(enable_timing=True); e_done.record(cur)
    torch.cuda.synchronize()
    p1, g1 = read()
    t_enc_conc = ev0.elapsed_time(e_done) / 4
    bw_beside = n_cov * bytes_per_add / (ev0.elapsed_time(ev1) * 1e-3) / 1e12
print('lib %s  B=%d' % (os.environ.get('TEPOSE_AMD_LIB', 'shipped'), B))
print('  alone : projection %.3f ms | recurrent %.3f ms | encoder %.2f ms | element-wise stream %.2f TB/s' % (p0, g0, t_enc, bw_alone))
print('  beside: projection %.3f ms | recurrent %.3f ms | encoder ~%.2f ms | element-wise stream %.2f TB/s (whole overlap, %d adds)'
      % (p1, g1, t_enc_conc, bw_beside, n_cov))
print('  => a split-off update moving 81 direction steps x 447 MB = 36.2 GB at the beside-rate would take %.2f ms next to a K-loop-only recurrent part of %.2f ms'
      % (36.2e9 / (bw_beside * 1e12) * 1e3, g1))
