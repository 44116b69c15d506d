"""Long-horizon check on one GPU: N steps of a 256 x 256 perturbed grid (the bench recipe, dt = 1e-7)
with the strict kernel, the fast kernel and the CPU oracle in lock-step (EDGE = edge of the 2-D grid;
SHAPE3 = "nx,ny,nz" runs the 3-D recipe instead).  Reports, every `every`
steps, the largest field difference strict-vs-oracle (expected: exactly 0), fast-vs-strict and, as
a yardstick for how the scheme itself amplifies a last-bit difference, the oracle with libm pow
against the oracle with T*sqrt(T) (all relative to the field scale), and the total-density
histories.  Writes one JSON document."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # tests/ is one level below the repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
from helpers import field_errors, oracle_params, bits_equal
from oracle import OracleB

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
edge = int(os.environ.get("EDGE", 256))
shape3 = tuple(int(v) for v in os.environ["SHAPE3"].split(",")) if os.environ.get("SHAPE3") else None
if shape3:
    params, fields, active = cases.synthetic_3d(*shape3, seed=4)
else:
    params, fields, active = cases.synthetic_2d(edge, edge, seed=4)
params["dt"] = 1e-7
ps, pf = dict(params, flags=0), dict(params, flags=4)
gs, gf = AdjGraph(ps, fields, active), AdjGraph(pf, fields, active)
ob = OracleB(oracle_params(params), fields, active, pow_mode=1, nthreads=min(16, os.cpu_count() or 1))
# yardstick: the oracle against itself with mu's T**1.5 through libm pow instead of T*sqrt(T) (a last-bit difference)
op = OracleB(oracle_params(params), fields, active, pow_mode=0, nthreads=min(16, os.cpu_count() or 1))
rows = []
t0 = time.time()
for s in range(0, nsteps, every):
    gs.step(every); gf.step(every); ob.step(every); op.step(every)
    fs, ff, fo = gs.read_fields(0), gf.read_fields(0), ob.read_fields(0)
    exact = all(bits_equal(fs[k], fo[k]) for k in range(7))
    d_so = max(field_errors(fs, fo))
    d_fs = max(field_errors(ff, fs))
    d_pw = max(field_errors(op.read_fields(0), fo))
    row = {"step": s + every, "oracle_pow_vs_sqrt": d_pw, "strict_equals_oracle_bitwise": bool(exact), "strict_vs_oracle": d_so, "fast_vs_strict": d_fs,
           "total_density_oracle": ob.total_density(), "total_density_strict": gs.total_density(),
           "total_density_fast": gf.total_density(), "finite": bool(np.all(np.isfinite(ff[0])))}
    rows.append(row)
    print(json.dumps(row), flush=True)
doc = {"grid": "%dx%dx%d" % shape3 if shape3 else "%dx%dx1" % (edge, edge),
       "recipe": "cases.synthetic_%s seed 4, dt 1e-7" % ("3d" if shape3 else "2d"), "steps": nsteps,
       "sweeps": 3 * nsteps, "seconds": time.time() - t0, "history": rows}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(doc, open(os.path.join(ROOT, "gpurun_out", "drift.json"), "w"), indent=1)
