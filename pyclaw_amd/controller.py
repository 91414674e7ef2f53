r"""
Minimal run loop (reference: src/pyclaw/controller.py:195-303).  Only what the solver
path needs to be driven like the reference's scripts do: output-time grid, ``keep_copy``
frames, ``solver.setup`` / ``evolve_to_time`` / ``teardown``.  File output, plotting and
restart (controller.py:77-130,307-433; src/pyclaw/io) are out of scope.
"""
import copy
import logging

import numpy as np


class Controller(object):
    def __init__(self):
        self.verbosity = 0
        self.solution = None
        self.solver = None
        self.keep_copy = False
        self.frames = []
        self.outstyle = 1
        self.nout = 10
        self.out_times = np.linspace(0.0, 1.0, self.nout + 1)
        self.nstepout = 1
        self.tfinal = 1.0
        self.output_format = None
        self.outdir = './_output'
        self.t0 = None
        self.logger = logging.getLogger('controller')

    def check_validity(self):
        if self.solver is None:
            raise Exception("No solver set in controller.")
        if self.solution is None:
            raise Exception("No solution set in controller.")
        if not self.solver.is_valid():
            raise Exception("The solver failed to initialize properly.")
        if not self.solution.is_valid():
            raise Exception("Initial solution is not valid.")

    def run(self):
        r"""controller.py:195-303 (outstyle 1, 2 and 3; no file output)."""
        frame = 0
        self.solver.setup(self.solution)
        self.solver.dt = self.solver.dt_initial
        self.check_validity()

        if self.outstyle == 1:
            output_times = np.linspace(self.solution.t, self.tfinal, self.nout + 1)
        elif self.outstyle == 2:
            output_times = self.out_times
        elif self.outstyle == 3:
            output_times = np.ones((self.nout + 1))
        else:
            raise Exception("Invalid output style %s" % self.outstyle)

        if self.keep_copy:
            self.frames.append(copy.deepcopy(self.solution))

        status = None
        for t in output_times[1:]:
            if self.outstyle < 3:
                status = self.solver.evolve_to_time(self.solution, t)
            else:
                for n in range(self.nstepout):
                    status = self.solver.evolve_to_time(self.solution)
            frame += 1
            if self.keep_copy:
                self.frames.append(copy.deepcopy(self.solution))
        self.solver.teardown()
        return status
