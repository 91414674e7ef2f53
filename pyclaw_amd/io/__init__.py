"""Frame output: the Clawpack ASCII format (reference: src/pyclaw/io/ascii.py) and block checkpoints."""
from .ascii import read_ascii, read_ascii_t, write_ascii
from .block import read_block, write_block

__all__ = ['read_ascii', 'read_ascii_t', 'write_ascii', 'read_block', 'write_block']
