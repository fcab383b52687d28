"""Drop-in for the two functions of the reference's lib/utils/geometry.py that sit on the hot path and that callers
also use on their own (lib/utils/demo_utils.py:112, lib/data_utils/threedpw_utils.py:98): device tensors in, device
tensors out, arithmetic in libtepose_hip.so (the device functions the regressor kernel inlines)."""
import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def rotation_matrix_to_angle_axis(rotation_matrix):
    """lib/utils/geometry.py:68-117 (+ :118-233): [N,3,3] (or the [N,3,4] the kornia-style signature allows; only the
    rotation part is read, as in the reference's own [N,3,3] branch) -> [N,3]; NaN -> 0."""
    R = rotation_matrix
    if not R.is_cuda:
        raise RuntimeError('tepose_amd runs on MI355X only: pass a cuda tensor (there is no CPU path)')
    if R.dim() != 3 or R.shape[1] != 3 or R.shape[2] not in (3, 4):
        raise ValueError('Input size must be a N x 3 x 3 (or N x 3 x 4) tensor; got {}'.format(tuple(R.shape)))
    R = R[:, :, :3].float().contiguous()
    aa = torch.empty((R.shape[0], 3), dtype=torch.float32, device=R.device)
    if R.shape[0]:
        with torch.cuda.device(R.device):
            _lib.check(_lib.load().tepose_rotmat_to_angle_axis(R.data_ptr(), R.shape[0], aa.data_ptr(), _stream()),
                       'tepose_rotmat_to_angle_axis')
    return aa


def rot6d_to_rotmat(x):
    """lib/utils/geometry.py:330-344: x viewed as [-1,3,2] -> [-1,3,3]."""
    if not x.is_cuda:
        raise RuntimeError('tepose_amd runs on MI355X only: pass a cuda tensor (there is no CPU path)')
    x6 = x.float().contiguous().view(-1, 6)
    R = torch.empty((x6.shape[0], 3, 3), dtype=torch.float32, device=x.device)
    if x6.shape[0]:
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().tepose_rot6d_to_rotmat(x6.data_ptr(), x6.shape[0], R.data_ptr(), _stream()),
                       'tepose_rot6d_to_rotmat')
    return R
