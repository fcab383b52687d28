"""`lib.models` shim, path-extending (see `lib/__init__.py`): `from lib.models import TePose`
(evaluate.py:15, train.py:19) resolves to the MI355X implementation, `lib.models.{tepose,spin,smpl,vibe}`
to the modules next to this file, and every other `lib.models.*` submodule (`motion_discriminator_gcn`,
`ms_gcn`, `resnet`, ...) to the caller's checkout.  `MotionDiscriminatorGCN` (training only,
lib/models/__init__.py:2) is imported from the checkout on first use, not eagerly: it drags in lib.graph."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from tepose_amd.tepose import TePose  # noqa: E402,F401  (lib/models/__init__.py:1)


def __getattr__(name):
    if name == 'MotionDiscriminatorGCN':
        from importlib import import_module
        return import_module('lib.models.motion_discriminator_gcn').MotionDiscriminatorGCN
    raise AttributeError('module %r has no attribute %r' % (__name__, name))
