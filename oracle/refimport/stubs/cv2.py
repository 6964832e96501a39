"""Stub for the reference's `import cv2` (utils/rend_util.py:1, render.py:16). Nothing on the
render hot path calls into it."""


def _unavailable(*a, **k):
    raise NotImplementedError("cv2 stub: not available in the oracle harness")


decomposeProjectionMatrix = imwrite = imread = _unavailable
