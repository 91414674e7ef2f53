"""Frame output in the Clawpack ASCII format (reference: src/pyclaw/io/ascii.py)."""
from .ascii import read_ascii, read_ascii_t, write_ascii

__all__ = ['read_ascii', 'read_ascii_t', 'write_ascii']
