"""
pyclaw_amd -- MI355X-native implementation of PyClaw's data-parallel hot path.

Same class names and solver surface as the reference package (``import pyclaw_amd as pyclaw``):
Dimension, Grid, State, Solution, BC, ClawSolver1D/2D/3D, SharpClawSolver1D/2D, Controller.  The classic
wave-propagation step runs in hand-written HIP kernels (libpyclaw_amd.so, include/pyclaw_amd.h);
there is no CPU fallback.
"""
from . import limiters, riemann
from .cfl import CFL
from .clawpack import ClawSolver1D, ClawSolver2D, ClawSolver3D, DeviceSource, EulerRadialSource, SphereCoriolisSource
from .controller import Controller
from .grid import Dimension, Grid
from .sharpclaw import SharpClawSolver1D, SharpClawSolver2D, DeviceDqSource, EulerRadialDqSource
from .solution import Solution
from .solver import BC, ConstantStateBC, DeviceBC, SphereMirrorBC
from .state import State

__all__ = ['limiters', 'riemann', 'CFL', 'ClawSolver1D', 'ClawSolver2D', 'ClawSolver3D', 'DeviceSource', 'EulerRadialSource', 'SphereCoriolisSource', 'SphereMirrorBC',
           'Controller', 'SharpClawSolver1D', 'SharpClawSolver2D', 'DeviceDqSource', 'EulerRadialDqSource', 'Dimension', 'Grid', 'Solution', 'BC', 'ConstantStateBC', 'DeviceBC', 'State']
