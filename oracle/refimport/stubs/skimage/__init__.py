"""Stub for `import skimage` (utils/io_util.py:17)."""
