r"""
Solution: a list of states + grids with a common time (reference: src/pyclaw/solution.py).
``write`` / ``read`` dispatch to ``pyclaw_amd.io`` ('ascii' frames, 'block' checkpoints; solution.py:356-448).
"""
from .grid import Grid
from .state import State


class Solution(object):
    def __init__(self, *arg, **kargs):
        self.states = []
        self.grids = []
        if len(arg) == 1 and isinstance(arg[0], State):
            self.states.append(arg[0])
            self.grids.append(arg[0].grid)
        elif len(arg) == 1 and isinstance(arg[0], (list, tuple)) and all(isinstance(s, State) for s in arg[0]):
            for s in arg[0]:
                self.states.append(s)
                self.grids.append(s.grid)
        elif len(arg) == 1 and isinstance(arg[0], Grid):
            raise Exception("A Solution is built from State objects: Solution(State(grid,meqn))")
        elif len(arg) >= 1 and isinstance(arg[0], int):
            # Solution(frame, path=..., format='ascii'): read a frame (solution.py:187-209)
            self.read(arg[0], kargs.get('path', './'), kargs.get('format', 'ascii'),
                      kargs.get('file_prefix', None), kargs.get('read_aux', False))
        elif len(arg) > 0:
            raise Exception("Invalid argument list")

    @property
    def state(self):
        return self.states[0]

    @property
    def grid(self):
        return self.grids[0]

    @property
    def t(self):
        return self.states[0].t

    @t.setter
    def t(self, value):
        for s in self.states:
            s.t = value

    @property
    def q(self):
        return self.states[0].q

    @property
    def aux(self):
        return self.states[0].aux

    def is_valid(self):
        return all(s.is_valid() for s in self.states)

    def write(self, frame, path='./', format='ascii', file_prefix=None, write_aux=False, options={},
              write_p=False):
        """solution.py:356-404 (ascii only)"""
        from . import io
        formats = format if isinstance(format, (list, tuple)) else [format]
        for fmt in formats:
            if fmt == 'ascii':
                io.write_ascii(self, frame, path, file_prefix or 'fort', write_aux, options, write_p)
            elif fmt == 'block':      # checkpoint: raw float64 block per rank + JSON header (io/block.py)
                io.write_block(self, frame, path, file_prefix or 'claw', write_aux, options, write_p)
            else:
                raise NotImplementedError("pyclaw_amd writes the formats 'ascii' and 'block'")

    def read(self, frame, path='./', format='ascii', file_prefix=None, read_aux=True, options={}):
        """solution.py:406-448 (ascii only)"""
        from . import io
        if format == 'ascii':
            io.read_ascii(self, frame, path, file_prefix or 'fort', read_aux, options)
        elif format == 'block':
            io.read_block(self, frame, path, file_prefix or 'claw', read_aux, options)
        else:
            raise NotImplementedError("pyclaw_amd reads the formats 'ascii' and 'block'")

    def __deepcopy__(self, memo={}):
        import copy
        result = self.__class__()
        for s in self.states:
            c = copy.deepcopy(s)
            result.states.append(c)
            result.grids.append(c.grid)
        return result
