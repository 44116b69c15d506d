"""Long-horizon run of the 8192 x 8192 grid (bench recipe, seed 4, dt = 1e-7): N steps of the default
(re-associated) kernels with the total density read back after every step, as a solver loop that monitors
conservation would.  Records the density history (every `every` steps), wall time and rate.
usage: long_horizon.py [steps] [every]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cfd_amd import cases
from cfd_amd.solver import AdjGraph

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 500
edge = int(os.environ.get("EDGE", 8192))
params, fields, active = cases.synthetic_2d(edge, edge, seed=4)
params["dt"] = 1e-7
params["flags"] = 4
g = AdjGraph(params, fields, active)
del fields
g.step(1)
d0 = g.total_density()
hist = [{"step": 1, "total_density": d0}]
t0 = time.perf_counter()
last = time.perf_counter()
for s in range(2, nsteps + 1):
    g.step(1)
    d = g.total_density()
    if s % every == 0:
        hist.append({"step": s, "total_density": d})
        print(json.dumps(hist[-1]), "%.1f s" % (time.perf_counter() - t0), flush=True)
elapsed = time.perf_counter() - t0
f = g.read_fields(0)
doc = {"grid": "%dx%dx1" % (edge, edge), "recipe": "cases.synthetic_2d seed 4, dt 1e-7, re-associated arithmetic",
       "steps": nsteps, "sweeps": 3 * nsteps, "density_reads": nsteps, "seconds": elapsed,
       "cell_updates_per_s": edge * edge * 3.0 * (nsteps - 1) / elapsed,
       "fields_finite_at_end": bool(all(np.all(np.isfinite(x)) for x in f)),
       "max_abs_U_at_end": float(np.max(np.abs(f[0]))), "max_abs_V_at_end": float(np.max(np.abs(f[1]))),
       "density_drift_rel": abs(hist[-1]["total_density"] - d0) / abs(d0), "history": hist}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(doc, open(os.path.join(ROOT, "gpurun_out", "long_horizon.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k != "history"}))
