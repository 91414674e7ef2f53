r"""
Grid data model: the boundary objects the solver surface takes (reference:
src/pyclaw/grid.py:36-145 Dimension, :149-363 Grid; parallel twin src/petclaw/grid.py:20-76).

Only what the solver path touches is restated: extents, cell counts, spacings,
centers/edges, and the per-process index range ``nstart/nend`` that the parallel layer
(pyclaw_amd/parallel.py) fills in when a grid is decomposed over GPUs.
"""
import numpy as np


class Dimension(object):
    r"""``Dimension(name, lower, upper, n)`` or ``Dimension(lower, upper, n)`` (grid.py:90-132)."""

    def __init__(self, *args, **kargs):
        self.name = 'x'
        self.n = None
        self.lower = 0.0
        self.upper = 1.0
        self.units = None
        if isinstance(args[0], float):
            self.lower = float(args[0])
            self.upper = float(args[1])
            self.n = int(args[2])
        elif isinstance(args[0], str):
            self.name = args[0]
            self.lower = float(args[1])
            self.upper = float(args[2])
            self.n = int(args[3])
        else:
            raise Exception("Invalid initializer for Dimension.")
        for (k, v) in kargs.items():
            setattr(self, k, v)
        # whole grid on this process until the parallel layer says otherwise
        self.nstart = 0
        self.nend = self.n
        self.lowerg = self.lower
        self._edge = None
        self._center = None

    @property
    def ng(self):
        """Number of cells of this dimension owned by this process (petclaw/grid.py:43-48)."""
        return self.nend - self.nstart

    @property
    def d(self):
        return (self.upper - self.lower) / float(self.n)

    @property
    def edge(self):
        if self._edge is None:
            self._edge = np.empty(self.ng + 1)
            for i in range(self.nstart, self.nend + 1):
                self._edge[i - self.nstart] = self.lower + i * self.d
        return self._edge

    @property
    def center(self):
        if self._center is None:
            self._center = np.empty(self.ng)
            for i in range(self.nstart, self.nend):
                self._center[i - self.nstart] = self.lower + (i + 0.5) * self.d
        return self._center

    def _set_range(self, nstart, nend):
        self.nstart = int(nstart)
        self.nend = int(nend)
        self.lowerg = self.lower + self.nstart * self.d
        self._edge = None
        self._center = None

    def __str__(self):
        output = "Dimension %s" % self.name
        if self.units:
            output += " (%s)" % self.units
        output += ":  (n,d,[lower,upper]) = (%s,%s,[%s,%s])" % (self.n, self.d, self.lower, self.upper)
        return output


class Grid(object):
    r"""``Grid(dimensions)``: ordered collection of Dimension objects (grid.py:296-362)."""

    def __init__(self, dimensions):
        self.level = 1
        self.gridno = 1
        self.gauges = []
        self.gauge_files = []
        self.gauge_path = './_output/_gauges/'
        self._dimensions = []
        if isinstance(dimensions, Dimension):
            dimensions = [dimensions]
        for dim in dimensions:
            self.add_dimension(dim)

    def add_dimension(self, dimension):
        if dimension.name in self._dimensions:
            raise Exception('Unable to add dimension. A dimension of the same name: %s, already exists.'
                            % dimension.name)
        self._dimensions.append(dimension.name)
        setattr(self, dimension.name, dimension)

    def add_gauges(self, gauge_coords):
        r"""Grid indices + output files of the gauges that lie in this rank's block (grid.py:519-545;
        file name and the floor(x/d) index rule are the reference's)."""
        import os
        from math import floor
        os.makedirs(self.gauge_path, exist_ok=True)
        for gauge in gauge_coords:
            gauge_ind = [int(floor(gauge[n] / self.d[n])) for n in range(self.ndim)]
            if all(self.nstart[n] <= gauge_ind[n] < self.nend[n] for n in range(self.ndim)):
                gauge_ind = [gauge_ind[n] - self.nstart[n] for n in range(self.ndim)]
                gauge_path = self.gauge_path + 'gauge' + '_'.join(str(coord) for coord in gauge) + '.txt'
                if os.path.isfile(gauge_path):
                    os.remove(gauge_path)
                self.gauges.append(list(gauge_ind))
                self.gauge_files.append(open(gauge_path, 'a'))

    # computational cell-centre coordinates as ndim mesh arrays (grid.py:255-291, 456-486)
    _c_center = None

    def compute_c_center(self, recompute=False):
        if recompute or self._c_center is None:
            if self.ndim == 1:
                self._c_center = [self.dimensions[0].center]
            else:
                index = np.indices(self.ng)
                self._c_center = [c[index[i, ...]] for i, c in enumerate(self.get_dim_attribute('center'))]

    @property
    def c_center(self):
        self.compute_c_center()
        return self._c_center

    p_center = c_center          # identity mapping (grid.py:293: mapc2p defaults to the identity)

    def get_dim_attribute(self, attr):
        return [getattr(getattr(self, name), attr) for name in self._dimensions]

    ndim = property(lambda self: len(self._dimensions))
    dimensions = property(lambda self: [getattr(self, name) for name in self._dimensions])
    n = property(lambda self: self.get_dim_attribute('n'))
    ng = property(lambda self: self.get_dim_attribute('ng'))
    nstart = property(lambda self: self.get_dim_attribute('nstart'))
    nend = property(lambda self: self.get_dim_attribute('nend'))
    name = property(lambda self: self._dimensions)
    lower = property(lambda self: self.get_dim_attribute('lower'))
    lowerg = property(lambda self: self.get_dim_attribute('lowerg'))
    upper = property(lambda self: self.get_dim_attribute('upper'))
    d = property(lambda self: self.get_dim_attribute('d'))
    units = property(lambda self: self.get_dim_attribute('units'))
    center = property(lambda self: self.get_dim_attribute('center'))
    edge = property(lambda self: self.get_dim_attribute('edge'))

    def __str__(self):
        return "Grid %s:\n" % self.gridno + "\n".join(str(getattr(self, d)) for d in self._dimensions)
