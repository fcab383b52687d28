from tepose_amd.vibe import VIBE, TemporalEncoder  # noqa: F401  (evaluate.py:16, demo.py:39)
