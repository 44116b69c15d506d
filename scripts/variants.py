"""Times the sweep kernel for several launch shapes (tuning aid).  usage: variants.py v:chunk ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
nz = int(os.environ.get("NZ", 1))
nx = int(os.environ.get("NX", 4096 if nz == 1 else 256)); ny = int(os.environ.get("NY", 4096 if nz == 1 else 256))
params, fields, active = cases.synthetic_2d(nx, ny, seed=1) if nz == 1 else cases.synthetic_3d(nx, ny, nz, seed=1)
bpc = 81 if nz == 1 else 97
params["dt"] = 1e-7
for spec in sys.argv[1:]:
    v, ch = spec.split(":")
    os.environ["CFD_VARIANT"] = v; os.environ["CFD_CHUNK"] = ch
    fl = 0
    if v.startswith("f"): fl |= 4; v = v[1:]
    if v == "s": fl |= 1; v = "0"
    os.environ["CFD_VARIANT"] = v
    p = dict(params); p["flags"] = fl
    g = AdjGraph(p, fields, active)
    for _ in range(30): g.sweep(False)
    g.set_timing(True); g.sweep_timing()
    for _ in range(40): g.sweep(False)
    ms, n = g.sweep_timing()
    print("variant %s chunk %s: %.3f ms/sweep  %.2f Gcell/s  %.0f GB/s" % (spec.split(":")[0], ch, ms / n, nx * ny * nz / (ms / n) / 1e6, nx * ny * nz * bpc / (ms / n) / 1e6), flush=True)
    g.close()
