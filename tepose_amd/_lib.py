"""ctypes binding of libtepose_hip.so (C ABI in include/tepose_amd.h).

There is no CPU or PyTorch fallback: if the shared library is missing the import
fails with instructions, and every non-zero return code raises.
"""
import ctypes
import os
from ctypes import c_uint, POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = (os.environ.get("TEPOSE_AMD_LIB") or os.path.join(_HERE, 'libtepose_hip.so'))   # override: A/B builds

# every symbol include/tepose_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    'tepose_version', 'tepose_error_string', 'tepose_create', 'tepose_destroy',
    'tepose_packed_bytes', 'tepose_set_blob', 'tepose_adopt_blob', 'tepose_pack_encoder', 'tepose_pack_regressor',
    'tepose_pack_smpl', 'tepose_jreg_packed_bytes', 'tepose_pack_jreg', 'tepose_workspace_bytes',
    'tepose_encoder_fwd', 'tepose_regressor_fwd', 'tepose_forward', 'tepose_gemm_workspace_bytes',
    'tepose_gemm_f32', 'tepose_profile_enable', 'tepose_profile_read', 'tepose_create_vibe', 'tepose_create_vibe_ex', 'tepose_vibe_feature_dim',
    'tepose_pack_vibe_encoder', 'tepose_vibe_workspace_bytes', 'tepose_vibe_encoder_fwd',
    'tepose_metrics_joints', 'tepose_smpl_verts_from_theta', 'tepose_metrics_verts', 'tepose_smpl_fwd', 'tepose_filter_one_euro', 'tepose_filter_slerp', 'tepose_project_frames', 'tepose_project_frame_pair', 'tepose_window_step', 'tepose_forward_cached', 'tepose_profile_read_gru', 'tepose_profile_read_l1proj', 'tepose_gemm_h3_workspace_bytes', 'tepose_gemm_h3_f32',
    'tepose_regressor_fwd_init', 'tepose_rotmat_to_angle_axis', 'tepose_rot6d_to_rotmat',
    'tepose_project_frames_workspace_bytes', 'tepose_smpl_fwd_per_person', 'tepose_joints_from_verts',
    'tepose_status', 'tepose_forward_status', 'tepose_status_peek', 'tepose_fault_code', 'tepose_set_persistent', 'tepose_uses_persistent', 'tepose_build_info',
    'tepose_fp32_ranges', 'tepose_derive_planes', 'tepose_kernel_info', 'tepose_select_kernels', 'tepose_set_option', 'tepose_get_option', 'tepose_debug_set_test_fault', 'tepose_debug_kernel_errors',
]

_lib = None


class TeposeError(RuntimeError):
    pass


E_TIMEOUT = -5


class TeposeTimeout(TeposeError):
    """TEPOSE_E_TIMEOUT: a persistent small-batch kernel gave up waiting for its peers (GPU shared or CU-masked);
    the outputs of that forward are NaN.  Engine re-runs on the step-per-launch kernels where it owns the sync point."""


class TeposeKernelFault(TeposeError):
    """(A sibling of TeposeTimeout, not a subclass: the retry loops that catch TeposeTimeout -- run_clips, StreamSession -- must not re-run on it.)
    The fault channel reported code 4: a bounded LDS poll of the barrier-free projection kernel (csrc/gemm_h3s16c.hip) expired.  No wait in that
    kernel depends on another workgroup, so this is a kernel bug or a hardware fault -- not a shared / CU-masked GPU, and the persistent small-batch
    kernels have nothing to do with it.  The forward's outputs are invalid (NaN-poisoned); the handle stays as it is."""


def load():
    """dlopen the library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            'tepose_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no fallback path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    fp = c_void_p  # device pointers travel as integers
    lib.tepose_version.restype = c_int
    lib.tepose_error_string.restype = c_char_p
    lib.tepose_error_string.argtypes = [c_int]
    lib.tepose_create.argtypes = [c_int, c_int, POINTER(c_void_p)]
    lib.tepose_destroy.argtypes = [c_void_p]
    lib.tepose_destroy.restype = None
    lib.tepose_packed_bytes.argtypes = [c_void_p]
    lib.tepose_packed_bytes.restype = c_size_t
    lib.tepose_set_blob.argtypes = [c_void_p, fp, c_size_t]
    lib.tepose_adopt_blob.argtypes = [c_void_p]
    lib.tepose_pack_encoder.argtypes = [c_void_p, POINTER(c_void_p), c_int, c_void_p]
    lib.tepose_pack_regressor.argtypes = [c_void_p, POINTER(c_void_p), c_int, c_void_p]
    lib.tepose_pack_smpl.argtypes = [c_void_p, fp, fp, fp, fp, fp, fp, POINTER(c_int32), c_void_p]
    lib.tepose_jreg_packed_bytes.restype = c_size_t
    lib.tepose_pack_jreg.argtypes = [fp, fp, c_void_p]
    lib.tepose_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.tepose_workspace_bytes.restype = c_size_t
    lib.tepose_encoder_fwd.argtypes = [c_void_p, fp, c_int, c_int, c_int, fp, fp, c_size_t, c_void_p]
    lib.tepose_regressor_fwd.argtypes = [c_void_p, fp, c_int, c_int, fp, fp, fp, fp, fp, fp, fp,
                                         c_size_t, c_void_p]
    lib.tepose_forward.argtypes = [c_void_p, fp, c_int, c_int, fp, fp, fp, fp, fp, fp, fp, c_size_t,
                                   c_void_p]
    lib.tepose_regressor_fwd_init.argtypes = [c_void_p, fp, c_int, c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp,
                                              c_size_t, c_void_p]
    lib.tepose_rotmat_to_angle_axis.argtypes = [fp, c_int, fp, c_void_p]
    lib.tepose_rot6d_to_rotmat.argtypes = [fp, c_int, fp, c_void_p]
    lib.tepose_gemm_workspace_bytes.argtypes = [c_int, c_int]
    lib.tepose_gemm_workspace_bytes.restype = c_size_t
    lib.tepose_gemm_f32.argtypes = [fp, c_long, fp, c_long, fp, fp, c_long, c_int, c_int, c_int, c_int,
                                    fp, c_size_t, c_void_p]
    lib.tepose_create_vibe.argtypes = [c_int, c_int, POINTER(c_void_p)]
    lib.tepose_create_vibe_ex.argtypes = [c_int, c_int, c_int, c_int, POINTER(c_void_p)]
    lib.tepose_vibe_feature_dim.argtypes = [c_void_p]
    lib.tepose_pack_vibe_encoder.argtypes = [c_void_p, POINTER(c_void_p), c_int, c_void_p]
    lib.tepose_vibe_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.tepose_vibe_workspace_bytes.restype = c_size_t
    lib.tepose_vibe_encoder_fwd.argtypes = [c_void_p, fp, c_int, c_int, c_int, fp, fp, c_size_t, c_void_p]
    lib.tepose_metrics_joints.argtypes = [fp, fp, c_int, c_int, c_int, fp, fp, fp, c_void_p]
    lib.tepose_smpl_verts_from_theta.argtypes = [c_void_p, fp, c_int, fp, fp, c_size_t, c_void_p]
    lib.tepose_metrics_verts.argtypes = [fp, fp, c_int, fp, c_void_p]
    lib.tepose_smpl_fwd.argtypes = [c_void_p, c_int, fp, fp, c_int, fp, fp, fp, c_size_t, c_void_p]
    lib.tepose_smpl_fwd_per_person.argtypes = [c_void_p, fp, fp, c_int, fp, fp, c_size_t, c_void_p]
    lib.tepose_filter_one_euro.argtypes = [fp, c_int, c_int, c_float, c_float, c_float, c_void_p]
    lib.tepose_filter_slerp.argtypes = [fp, fp, c_int, c_int, c_double, c_void_p]
    lib.tepose_project_frames_workspace_bytes.argtypes = [c_void_p, c_int]
    lib.tepose_project_frames_workspace_bytes.restype = c_size_t
    lib.tepose_project_frames.argtypes = [c_void_p, fp, c_long, fp, c_long, c_int, fp, c_long, fp, c_size_t, c_void_p]
    lib.tepose_project_frame_pair.argtypes = [c_void_p, fp, fp, c_long, fp, c_long, c_int, fp, c_long, fp, c_long, fp, c_size_t, c_void_p]
    lib.tepose_window_step.argtypes = [c_void_p, fp, fp, c_long, fp, c_long, fp, c_long, fp, c_long, fp, c_int, c_int, c_long, c_int, c_int, fp, fp, fp, fp, fp, fp,
                                       fp, c_size_t, fp, c_size_t, c_void_p]
    lib.tepose_forward_cached.argtypes = [c_void_p, fp, c_int, c_int, c_long, fp, c_long, c_int, c_int, fp, fp, fp, fp,
                                          fp, fp, fp, c_size_t, c_void_p]
    lib.tepose_gemm_h3_workspace_bytes.argtypes = [c_int, c_int, c_int]
    lib.tepose_gemm_h3_workspace_bytes.restype = c_size_t
    lib.tepose_gemm_h3_f32.argtypes = [fp, c_long, fp, c_long, fp, fp, c_long, c_int, c_int, c_int, fp, c_size_t, c_void_p]
    lib.tepose_fp32_ranges.argtypes = [c_void_p, POINTER(c_size_t), POINTER(c_size_t), c_int]
    lib.tepose_derive_planes.argtypes = [c_void_p, c_void_p]
    lib.tepose_status.argtypes = [c_void_p, c_void_p]
    lib.tepose_forward_status.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.tepose_debug_set_test_fault.argtypes = [c_void_p, c_uint]
    lib.tepose_debug_kernel_errors.restype = c_uint
    lib.tepose_status_peek.argtypes = [c_void_p]
    lib.tepose_fault_code.argtypes = [c_void_p]
    lib.tepose_set_persistent.argtypes = [c_void_p, c_int]
    lib.tepose_uses_persistent.argtypes = [c_void_p, c_int, c_int]
    lib.tepose_joints_from_verts.argtypes = [c_void_p, fp, fp, c_int, fp, c_void_p]
    lib.tepose_profile_enable.argtypes = [c_void_p, c_int]
    lib.tepose_profile_read.argtypes = [c_void_p, POINTER(c_double), POINTER(c_int), POINTER(c_double)]
    lib.tepose_profile_read_gru.argtypes = [c_void_p, POINTER(c_double), POINTER(c_int), POINTER(c_double)]
    lib.tepose_profile_read_l1proj.argtypes = [c_void_p, POINTER(c_double), POINTER(c_int), POINTER(c_double)]
    for name in SYMBOLS:
        getattr(lib, name)              # AttributeError here = the built library is older than this binding
    if lib.tepose_version() != 1:
        raise ImportError('tepose_amd: ABI version mismatch (%d)' % lib.tepose_version())
    lib.tepose_build_info.restype = c_char_p
    lib.tepose_kernel_info.restype = c_char_p
    lib.tepose_kernel_info.argtypes = [c_void_p]
    lib.tepose_select_kernels.restype = c_char_p
    lib.tepose_set_option.argtypes = [c_void_p, c_char_p, ctypes.c_long]
    lib.tepose_get_option.argtypes = [c_void_p, c_char_p]
    lib.tepose_get_option.restype = ctypes.c_long
    lib.tepose_select_kernels.argtypes = [c_void_p, c_int, c_int]
    info = (lib.tepose_build_info() or b'').decode()
    if 'packed_fp32=off' not in info:
        import warnings
        warnings.warn('tepose_amd: %s was built WITH packed-fp32 VALU instructions (%r); on a GPU shared by two processes such a '
                      'build returned wrong SMPL vertices in 1-3 %% of the launches (DESIGN.md section 10) -- rebuild with '
                      '__graft_entry__.build()' % (LIB_PATH, info), RuntimeWarning)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().tepose_error_string(rc)
        raise (TeposeTimeout if rc == E_TIMEOUT else TeposeError)('%s failed: %s (code %d)' % (what, msg.decode() if msg else '?', rc))


def ptr_array(ptrs):
    arr = (c_void_p * len(ptrs))(*ptrs)
    return arr
