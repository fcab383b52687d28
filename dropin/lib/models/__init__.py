"""`lib.models` shim: put `<repo>/dropin` (and `<repo>`) ahead of the reference checkout on
sys.path and `from lib.models import TePose` (evaluate.py:15, train.py:19) resolves to the
MI355X implementation.  MotionDiscriminatorGCN (training only) is deliberately not exported."""
from tepose_amd.tepose import TePose, TemporalEncoder  # noqa: F401
