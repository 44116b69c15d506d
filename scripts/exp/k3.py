"""Interleaved timing of 3-D sweep launch shapes in ONE process (same GPU, alternating rounds).
usage: k3.py spec...   spec = variant:chunk  (fast arithmetic); env N = cube edge (256), ROUNDS (5), LIBS = name=path,... extra builds
Prints per spec the median / min kernel time of the plain sweep and of the sweep fused with the filter."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
n = int(os.environ.get("N", 256)); rounds = int(os.environ.get("ROUNDS", 5)); reps = int(os.environ.get("REPS", 6))
params, fields, active = cases.synthetic_3d(n, n, n, seed=3)
params["dt"] = 1e-7; params["flags"] = 4
gs = []
for spec in sys.argv[1:]:
    v, ch = spec.split(":")[:2]
    os.environ["CFD_VARIANT"] = v; os.environ["CFD_CHUNK"] = ch
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
    print("%-10s %s  plain median %.4f min %.4f ms | fused median %.4f min %.4f ms | %.1f Gcell/s plain" % (
        spec, g.kernel_name(), statistics.median(tp[1:]), min(tp), statistics.median(tf[1:]), min(tf),
        n ** 3 / statistics.median(tp[1:]) / 1e6), flush=True)
