r"""
Limiter ids of ``solver.limiters`` / ``mthlim`` (reference: src/pyclaw/limiters/tvd.py:74-78,
philim.f:19-55).  The ids the device ``philim`` implements are 0-5; the reference's Python-only
limiters (ids > 5) have no Fortran counterpart and are not part of the kernels.
"""
minmod = 1
superbee = 2
vanleer = 3
MC = 4
Beam_Warming = 5
