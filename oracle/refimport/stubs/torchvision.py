"""Stub for `import torchvision` (utils/logger.py:8)."""
