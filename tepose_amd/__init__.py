"""tepose_amd -- MI355X (gfx950) implementation of the TePose per-window inference hot path.

Public names mirror the reference's lib.models.  Importing this package
does not load the HIP library; constructing a model does, and fails loudly if it is missing.
"""
from .smpl import SMPL, SMPL_MODEL_DIR, SMPL_MEAN_PARAMS, H36M_TO_J14, JOINT_MAP, JOINT_NAMES  # noqa: F401
from .spin import Regressor, projection  # noqa: F401
from .tepose import TePose, TemporalEncoder  # noqa: F401

__all__ = ['TePose', 'TemporalEncoder', 'Regressor', 'projection', 'SMPL', 'SMPL_MODEL_DIR',
           'SMPL_MEAN_PARAMS', 'H36M_TO_J14', 'JOINT_MAP', 'JOINT_NAMES']
