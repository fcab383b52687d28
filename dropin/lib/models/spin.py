from tepose_amd.spin import Regressor, projection, hmr  # noqa: F401  (vibe.py:24, demo.py:116)
