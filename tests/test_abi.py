"""CPU: the C-ABI library builds/loads and exports every symbol include/tepose_amd.h declares;
host-side argument checks that need no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from tepose_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    from tepose_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'tepose_amd.h')).read()
    declared = set(re.findall(r'\b(tepose_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_errors(lib):
    assert lib.tepose_version() == 1
    assert b'workspace' in lib.tepose_error_string(-3)


def test_handle_argument_errors_without_gpu(lib):
    h = ctypes.c_void_p()
    assert lib.tepose_create(0, 1024, ctypes.byref(h)) == -1
    assert lib.tepose_create(2, 1024, ctypes.byref(h)) == 0
    n = lib.tepose_packed_bytes(h)
    # encoder 60.6M + regressor 3.5M + SMPL ~5M floats (padded) + 54M floats of fp16 hi/lo planes of the
    # GRU matrices for the split-precision GEMM
    assert 700e6 < n < 840e6
    # nothing packed yet -> state error, before any device access
    assert lib.tepose_encoder_fwd(h, 16, 1, 1, 0, 16, 16, 1 << 20, None) == -4
    assert lib.tepose_workspace_bytes(h, 64, 16) > 0
    assert lib.tepose_workspace_bytes(h, 0, 16) == 0
    lib.tepose_destroy(h)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tepose_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(ImportError, match='no fallback'):
        _lib.load()


def test_cpu_input_is_rejected():
    import torch
    from tepose_amd.engine import check_input
    with pytest.raises(RuntimeError, match='no CPU path'):
        check_input(torch.zeros(1, 6, 2133))
    with pytest.raises(ValueError):
        check_input(torch.zeros(1, 6, 2048))


def test_state_dict_keys_match_reference_layout():
    """Appendix B of SURVEY.md: same key names/shapes as the reference model (L=2, H=1024)."""
    from tepose_amd import synth
    from tepose_amd.testing import build_model
    model, state, _ = build_model(1, 64, device='cpu')
    sd = model.state_dict()
    for k, shape in synth.encoder_param_shapes(1, 64) + synth.regressor_param_shapes():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert 'regressor.smpl.J_regressor_extra' in sd
    # a checkpoint with unknown smplx-owned keys still loads strictly (evaluate.py:124)
    import torch
    sd2 = dict(sd)
    sd2['regressor.smpl.some_future_smplx_buffer'] = torch.zeros(3)
    sd2['regressor.smpl.betas'] = torch.zeros(7, 10)       # other batch size
    model.load_state_dict(sd2, strict=True)
