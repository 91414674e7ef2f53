"""Limiter ids (reference: src/pyclaw/limiters/tvd.py); the limiting itself runs in the HIP kernels."""
from . import tvd

__all__ = ['tvd']
