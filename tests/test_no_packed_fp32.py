"""CPU: the shipped libtepose_hip.so contains NO packed-fp32 VALU instruction.

gfx950 erratum found in round 4 (reproducer tools/micro/pk_chain_mfma.hip, 120 lines, one process):
`v_pk_fma_f32 ... op_sel:[0,1,0]` -- the form whose LOW result lane reads the HIGH half of src1, which hipcc emits when it packs
row-times-vector code such as the skinning kernel's `t0 x + t1 y + t2 z + t3` -- returns a wrong low result in lanes 48..63
while other workgroups on the same CU run ds_read_b128 + v_mfma loops (i.e. next to any of this library's GEMM kernels, in
another process or on another stream).  The library is therefore compiled with `-Xclang -target-feature -Xclang
-packed-fp32-ops` (__graft_entry__.HIPCC_EXTRA); this test disassembles every gfx950 code object of the built library and
fails if a v_pk_{fma,mul,add}_f32 slipped back in (a new source file built without the flag, a dropped flag, ...)."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


@pytest.mark.skipif(not os.path.isfile(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_shipped_library_has_no_packed_fp32_instructions(tmp_path):
    import __graft_entry__ as g
    g.build()
    so = tmp_path / 'libtepose_hip.so'
    shutil.copy(g.LIB, so)
    p = subprocess.run([OBJDUMP, '--offloading', str(so)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    objs = sorted(glob.glob(str(tmp_path / 'libtepose_hip.so.*gfx950*')))
    assert len(objs) >= len(g.SOURCES) - 1, objs                 # one device code object per translation unit that has kernels
    packed = re.compile(r'\bv_pk_(fma|mul|add)_f32\b')
    total = mfma = 0
    for o in objs:
        d = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', o], capture_output=True, text=True, timeout=600)
        assert d.returncode == 0, d.stderr[-2000:]
        hits = [l.strip() for l in d.stdout.splitlines() if packed.search(l)]
        assert not hits, '%s: %d packed-fp32 instructions, e.g. %s' % (os.path.basename(o), len(hits), hits[:3])
        total += d.stdout.count('\n')
        mfma += d.stdout.count('v_mfma_')
    assert total > 50000 and mfma > 1000                          # the disassembly really is the library's device code
    info = subprocess.run(['python', '-c', 'from tepose_amd import _lib; print(_lib.load().tepose_build_info().decode())'],
                          cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert 'packed_fp32=off' in info.stdout, info.stdout + info.stderr
