import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import pyclaw_amd as pyclaw
from apps import problems
from oracle import driver as D, oracle as O
claw = problems.shockbubble(pyclaw, tfinal=0.2)
p = D.shockbubble_problem(); D.run(p, O.COracle(), 0.2, 1)
q = claw.frames[1].state.q
for m in range(5):
    d = np.abs(q[m]-p.q[m]); print(m, d.max(), np.argwhere(d>0)[:4].tolist(), (d>0).sum())
print(claw.solver.status, )
