"""`lib` shim, path-extending: `<repo>/dropin` goes AHEAD of the caller's TePose checkout on sys.path.

Only `lib.models.{tepose,spin,smpl,vibe}` live here; every other `lib.*` module (`lib.core.config`,
`lib.utils.*`, `lib.data_utils.*`, `lib.dataset.*`, `lib.graph.*` ...) keeps resolving to the checkout,
because `__path__` is extended with every other `lib/` directory on sys.path (the reference's `lib/`
has no `__init__.py`, i.e. it is a namespace portion: `pkgutil.extend_path` picks up both kinds).
So `evaluate.py:12-21` / `demo.py:18-41` import unchanged (tests/test_dropin.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
