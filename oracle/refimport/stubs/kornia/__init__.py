"""Stub for `kornia` (utils/metric_util.py:3)."""
