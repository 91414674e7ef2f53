r"""
Solution: a list of states + grids with a common time (reference: src/pyclaw/solution.py).
Frame reading/writing (solution.py:356-448, src/pyclaw/io) is out of scope.
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
        elif len(arg) > 0:
            raise Exception("Invalid argument list; frame reading is not part of pyclaw_amd")

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

    def __deepcopy__(self, memo={}):
        import copy
        result = self.__class__()
        for s in self.states:
            c = copy.deepcopy(s)
            result.states.append(c)
            result.grids.append(c.grid)
        return result
