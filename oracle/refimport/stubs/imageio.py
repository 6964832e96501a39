"""Stub for `import imageio` (utils/io_util.py:9, utils/logger.py:7, render.py:12)."""


def _unavailable(*a, **k):
    raise NotImplementedError("imageio stub")


imread = imwrite = mimwrite = _unavailable
