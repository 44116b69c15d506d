"""Interleaved timing of strict (reference operation order) sweep launch shapes in ONE process.
usage: kstrict.py spec...   spec = variant:chunk (chunk 'auto' = the library's choice); env DIM = 2|3 (3), N = edge (256 in 3-D, 4096 in 2-D), ROUNDS (4)
Prints per spec the median / min time of the plain sweep and of the sweep fused with the filter (per sweep,
all of its launches together)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
dim = int(os.environ.get("DIM", 3))
n = int(os.environ.get("N", 256 if dim == 3 else 4096)); rounds = int(os.environ.get("ROUNDS", 4)); reps = int(os.environ.get("REPS", 4))
if dim == 3:
    params, fields, active = cases.synthetic_3d(n, n, n, seed=3)
else:
    params, fields, active = cases.synthetic_2d(n, n, seed=1)
params["dt"] = 1e-7; params["flags"] = 0
cells = n ** dim
gs = []
for spec in sys.argv[1:]:
    v, ch = spec.split(":")[:2]
    os.environ["CFD_VARIANT"] = v
    if ch == "auto": os.environ.pop("CFD_CHUNK", None)     # the library's own choice
    else: os.environ["CFD_CHUNK"] = ch
    g = AdjGraph(params, fields, active)
    g.step(2)
    g.set_timing(True); g.sweep_timing(); g.kernel_timing(1)
    gs.append((spec, g, [], []))
for r in range(rounds):
    for spec, g, tp, tf in gs:
        for _ in range(reps): g.sweep(False)
        ms, k = g.sweep_timing(); tp.append(ms / k)
        for _ in range(reps): g.sweep_and_filter()
        ms, k = g.kernel_timing(1); tf.append(ms / max(k, 1))
for spec, g, tp, tf in gs:
    print("%-10s %-45s plain median %.4f min %.4f ms | fused median %.4f min %.4f ms | %.1f Gcell/s plain" % (
        spec, g.kernel_name(), statistics.median(tp[1:]), min(tp), statistics.median(tf[1:]), min(tf),
        cells / statistics.median(tp[1:]) / 1e6), flush=True)
