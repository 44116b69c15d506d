"""Profiling target: N sweeps of the 4096 x 4096 synthetic grid, or NX x NY x NZ with NZ > 1 (kernel selected by FLAGS / CFD_VARIANT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
nz = int(os.environ.get("NZ", 1))
nx = int(os.environ.get("NX", 4096 if nz == 1 else 256)); ny = int(os.environ.get("NY", 4096 if nz == 1 else 256))
params, fields, active = cases.synthetic_2d(nx, ny, seed=1) if nz == 1 else cases.synthetic_3d(nx, ny, nz, seed=3)
params["dt"] = 1e-7
params["flags"] = int(os.environ.get("FLAGS", "4"))
g = AdjGraph(params, fields, active)
for i in range(int(os.environ.get("NSWEEP", "6"))):
    g.sweep(i % 3 == 0)
g.time_filter()
g.step(int(os.environ.get("NSTEP", "3")))     # whole steps: the launch kinds the default path uses
g.sync()
print("done", g.kernel_name())
