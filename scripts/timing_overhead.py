"""Cost of the per-dispatch timing events bench.py switches on in its timed region: the default 4096 x 4096
workload stepped with and without them, alternating."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
params, fields, active = cases.synthetic_2d(4096, 4096, seed=1)
params["dt"] = 1e-7; params["flags"] = 4
g = AdjGraph(params, fields, active)
g.step(20); g.sync()
def run(timing, n=200):
    g.set_timing(timing); g.sweep_timing(reset=True)
    g.sync(); t0 = time.perf_counter()
    g.step(n)
    g.sync(); dt = (time.perf_counter() - t0) / n * 1e3
    g.sweep_timing(reset=True); g.kernel_timing(2, reset=True); g.set_timing(False)
    return dt
for rep in range(3):
    print("events off %.4f ms/step   events on %.4f ms/step" % (run(False), run(True)))
