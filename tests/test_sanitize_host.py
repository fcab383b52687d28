"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the host side of libtepose_hip.so.  tools/sanitize_host.sh builds the library
with -fsanitize=address,undefined on the host code (device code as shipped, -fno-gpu-sanitize) into build/san/, links the plain-C exerciser
tests/c_client/tepose_host_check.c (20 000 calls: layouts for ten (layers, hidden) pairs, sizes and kernel selection over the batch-class grid, the option table with
extreme values, every argument-error path) and runs it; the full script also runs tests/test_dispatch.py + tests/test_abi.py on the instrumented library
(TEPOSE_SANITIZE_FULL=1 here).  Skipped where the sanitizer runtime is missing; never run on the GPU pool."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')

pytestmark = pytest.mark.skipif(not RT or os.environ.get('TEPOSE_SANITIZE_RUN') == '1', reason='sanitizer runtime missing (or already inside a sanitizer run)')


def test_host_side_is_clean_under_asan_and_ubsan():
    full = os.environ.get('TEPOSE_SANITIZE_FULL') == '1'
    p = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'sanitize_host.sh')] + ([] if full else ['--quick']), cwd=ROOT, capture_output=True, text=True, timeout=1800)
    tail = (p.stdout + p.stderr)[-4000:]
    assert p.returncode == 0, tail
    assert 'host check ok' in p.stdout and 'sanitize_host: no report' in p.stdout, tail
    assert 'ERROR: AddressSanitizer' not in tail and 'runtime error' not in tail, tail
