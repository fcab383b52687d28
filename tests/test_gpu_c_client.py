"""GPU: the C ABI driven by a plain C program (tests/c_client/tepose_client.c: no Python, no torch, no C++ types in the
calls) reproduces the Python drop-in bit for bit -- the boundary a cgo / JNI / ctypes binding would use is complete."""
import os
import subprocess

import numpy as np
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _find_gcc_and_rocm():
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    if not os.path.exists(os.path.join(rocm, 'include', 'hip', 'hip_runtime_api.h')):
        pytest.skip('HIP runtime headers not found')
    return rocm


@pytest.mark.parametrize('L,H,B,T,use_jreg', [(1, 64, 2, 4, True), (2, 128, 5, 6, False)])
def test_plain_c_caller_matches_the_python_dropin(L, H, B, T, use_jreg, tmp_path):
    from tepose_amd import _lib
    from tepose_amd.testing import build_model
    rocm = _find_gcc_and_rocm()
    exe = str(tmp_path / 'tepose_client')
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ['gcc', '-O1', '-std=c99', '-D__HIP_PLATFORM_AMD__', '-I' + os.path.join(rocm, 'include'),
           os.path.join(ROOT, 'tests', 'c_client', 'tepose_client.c'), '-o', exe, '-L' + libdir, '-l:' + os.path.basename(_lib.LIB_PATH),
           '-L' + os.path.join(rocm, 'lib'), '-lamdhip64', '-Wl,-rpath,' + libdir, '-Wl,-rpath,' + os.path.join(rocm, 'lib')]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]

    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(L, H, seed=31, device='cuda', smpl_np=smpl_np)
    eng = model._engine
    x = torch.from_numpy(synth.synthetic_windows(B, T, 32)).cuda()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    with torch.no_grad():
        ref = model(x, J_regressor=J if use_jreg else None)[0]
    inp = str(tmp_path / 'in.bin')
    with open(inp, 'wb') as f:
        np.array([L, H, B, T, int(use_jreg)], dtype=np.int32).tofile(f)
        smpl = model.regressor.smpl
        for t in eng._enc_tensors(model.encoder) + eng._reg_tensors(model.regressor) + eng._smpl_tensors(smpl)[:6]:
            t.detach().float().cpu().contiguous().numpy().tofile(f)
        smpl.parents.detach().cpu().numpy().astype(np.int32).tofile(f)
        if use_jreg:
            J.float().contiguous().numpy().tofile(f)
        x.cpu().numpy().tofile(f)
    outp = str(tmp_path / 'out.bin')
    env = {k: v for k, v in os.environ.items() if not k.startswith('TEPOSE_')}
    p = subprocess.run([exe, inp, outp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert 'tepose_client ok' in p.stdout
    got = np.fromfile(outp, dtype=np.float32)
    nj = 14 if use_jreg else 49
    off = 0
    for key, shape in (('theta', (B, 85)), ('verts', (B, 6890, 3)), ('kp_3d', (B, nj, 3)), ('kp_2d', (B, nj, 2)),
                       ('rotmat', (B, 24, 3, 3))):
        n = int(np.prod(shape))
        a = got[off:off + n].reshape(shape)
        off += n
        assert np.array_equal(a, ref[key].cpu().numpy()), key          # same library, same kernels: same bits
    assert off == got.size
