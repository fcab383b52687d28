"""GPU: the row tile of gemm_h3_kernel is a scheduling choice (64-row tiles where they save rounds of the chip, launch_gemm_h3; profiles/r05_mid_rows_gemm.txt):
which workgroup owns an element changes, its K order does not -- results are bit-identical with the rules off (TEPOSE_H3_TILE64=0, TEPOSE_H3_TILE192=0 -- the
128 x 192 tiles of round 6, taken where they turn a round and a bit of 128 x 128 tiles into one round: 1536 x 3072 --; read at tepose_create / per call of the
handle-less entry: two subprocesses) and both agree with fp64."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from tepose_amd import _lib
lib = _lib.load()
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(5)
for M, N, K in ((150, 1152, 256), (444, 3072, 512), (64, 640, 96), (1030, 384, 160), (1536, 3072, 256), (1100, 1536, 128)):
    A = torch.randn(M, K, device=dev, generator=g)
    W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    C = torch.full((M, N), float('nan'), device=dev)
    ws = torch.empty(lib.tepose_gemm_h3_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev)
    assert lib.tepose_gemm_h3_f32(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, M, N, K, ws.data_ptr(), ws.numel(), st) == 0
    torch.cuda.synchronize()
    err = (C.double() - (A.double() @ W.double().t() + b.double())).abs().max().item()
    print(M, N, K, hashlib.sha256(C.cpu().numpy().tobytes()).hexdigest(), '%%.3e' %% err)
''' % ROOT


def run(env_value):
    env = dict(os.environ)
    env.pop('TEPOSE_H3S', None)
    env.pop('TEPOSE_H3_TILE', None)
    for k in ('TEPOSE_H3_TILE64', 'TEPOSE_H3_TILE192'):
        if env_value is None:
            env.pop(k, None)
        else:
            env[k] = env_value
    p = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return [l.split() for l in p.stdout.strip().splitlines() if l and l[0].isdigit()]


def test_tile_rule_is_bit_identical_to_128_row_tiles():
    a, b = run(None), run('0')
    assert len(a) == 6 and len(b) == 6
    for ra, rb in zip(a, b):
        assert ra[:4] == rb[:4], (ra, rb)                  # same shape, same bytes
        assert float(ra[4]) < 2e-5, ra
