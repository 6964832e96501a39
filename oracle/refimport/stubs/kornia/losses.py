"""Stub for `from kornia.losses import ssim` (utils/metric_util.py:3)."""


def ssim(*a, **k):
    raise NotImplementedError("kornia stub")
