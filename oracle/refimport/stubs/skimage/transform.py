"""Stub for `from skimage.transform import rescale` (utils/io_util.py:18)."""


def rescale(*a, **k):
    raise NotImplementedError("skimage stub")
