r"""
Running Courant number of a solver (the role of src/pyclaw/cfl.py:1-18 and, for decomposed
grids, src/petclaw/cfl.py:1-33).

The sweep kernels reduce the Courant number on the device (one u64 atomicMax per wavefront,
classic.hpp: cfl_publish) and, when the grid is split over several GPUs, all-reduce that word
with RCCL before it is read back (pclaw.hip: read_cfl).  What reaches this object is therefore
already the maximum over every cell of every block; it only has to remember the largest value
it was shown -- the three-method surface the reference's solvers call.
"""


class CFL(object):
    __slots__ = ('_largest', '_reduce')

    def __init__(self, global_max):
        self._largest = global_max
        self._reduce = None      # optional hook: value -> value (kept for host-side test transports)

    def get_cached_max(self):
        """Largest Courant number recorded since the last update_global_max."""
        return self._largest

    def get_global_max(self):
        return self._largest

    def set_local_max(self, new_local_max):
        self._largest = new_local_max if self._reduce is None else self._reduce(new_local_max)

    def update_global_max(self, new_local_max):
        self.set_local_max(new_local_max)
