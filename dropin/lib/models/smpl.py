from tepose_amd.smpl import *  # noqa: F401,F403
from tepose_amd.smpl import SMPL, SMPL_MODEL_DIR, SMPL_MEAN_PARAMS, H36M_TO_J14, JOINT_MAP, JOINT_NAMES, JOINT_IDS  # noqa: F401  (evaluate.py:17, demo.py:20)
