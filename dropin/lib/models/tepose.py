from tepose_amd.tepose import TePose, TemporalEncoder  # noqa: F401  (demo.py:38)
