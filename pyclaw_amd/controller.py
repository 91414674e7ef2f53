r"""
Run loop (reference: src/pyclaw/controller.py:195-303): output-time grid, ``keep_copy`` frames,
frame files in the Clawpack ASCII format (``output_format='ascii'``), output functionals
(``compute_F``/``write_F``, controller.py:307-317), ``solver.setup`` / ``evolve_to_time`` /
``teardown``.  Plotting and the hdf5/netcdf/petsc formats are out of scope.
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
        self.output_format = None       # 'ascii' writes fort.qNNNN / fort.tNNNN into outdir
        self.outdir = './_output'
        self.output_file_prefix = None
        self.output_options = {}
        self.write_aux_init = False
        self.write_aux_always = False
        self.compute_p = None
        self.compute_F = None
        self.F_file_name = 'F'
        self.start_frame = 0
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
        r"""controller.py:195-303 (outstyle 1, 2 and 3)."""
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
        frame = self.start_frame
        if self.output_format is not None:
            if self.compute_p is not None:
                self.compute_p(self.solution.state)
                self.solution.write(frame, self.outdir, self.output_format, self.output_file_prefix + '_p'
                                    if self.output_file_prefix else 'claw_p', write_aux=False, write_p=True)
            self.solution.write(frame, self.outdir, self.output_format, self.output_file_prefix,
                                self.write_aux_init, self.output_options)
        self.write_F('w')
        self.solver.write_gauge_values(self.solution)    # initial gauge values (controller.py:225-226)

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
            if self.output_format is not None:
                if self.compute_p is not None:
                    self.compute_p(self.solution.state)
                    self.solution.write(frame, self.outdir, self.output_format, self.output_file_prefix + '_p'
                                        if self.output_file_prefix else 'claw_p', write_aux=False, write_p=True)
                self.solution.write(frame, self.outdir, self.output_format, self.output_file_prefix,
                                    self.write_aux_always, self.output_options)
            self.write_F()
            for f in self.solution.state.grid.gauge_files:
                f.flush()
        self.solver.teardown()
        for f in self.solution.state.grid.gauge_files:
            f.close()
        return status

    def write_F(self, mode='a'):
        """Output functionals sum|F_i| at output times (controller.py:307-317; petclaw sums over ranks)."""
        if self.compute_F is None:
            return
        import os
        from . import parallel
        state = self.solution.state
        self.compute_F(state)
        sums = [float(np.sum(np.abs(state.F[i, ...]))) for i in range(state.F.shape[0])]
        sums = parallel.allreduce_sum_host(sums)
        if parallel.rank() == 0:
            os.makedirs(self.outdir, exist_ok=True)
            with open(os.path.join(self.outdir, self.F_file_name + '.txt'), mode) as f:
                f.write(' '.join([str(self.solution.t)] + [str(v) for v in sums]) + '\n')
