r"""
CFL bookkeeping object (reference: src/pyclaw/cfl.py:5-25; parallel twin src/petclaw/cfl.py:5-31).
"""


class CFL(object):
    def __init__(self, global_max):
        self._global_max = global_max
        self._reduce = None     # set by the parallel layer: callable(local_max) -> global max

    def get_global_max(self):
        return self._global_max

    def get_cached_max(self):
        return self._global_max

    def set_local_max(self, new_local_max):
        self._global_max = new_local_max

    def update_global_max(self, new_local_max):
        if self._reduce is not None:
            new_local_max = self._reduce(new_local_max)
        self._global_max = new_local_max
