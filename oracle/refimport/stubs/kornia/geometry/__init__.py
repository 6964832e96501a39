"""Stub package for `kornia.geometry` (editing/render_geometry_editing.py:8)."""
