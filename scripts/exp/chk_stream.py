import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
from helpers import field_errors, bits_equal
params, fields, active = cases.synthetic_2d(700, 500, seed=6)
def run(variant, chunk, steps=2):
    os.environ["CFD_VARIANT"] = str(variant); os.environ["CFD_CHUNK"] = str(chunk)
    p = dict(params); p["flags"] = 4
    g = AdjGraph(p, fields, active); g.step(steps)
    out = [g.read_fields(r) for r in range(3)]; td = g.total_density(); g.close(); return out, td
ref, tdr = run(8, 32)
for v, ch in [(20, 32), (20, 7), (21, 32), (22, 32), (23, 32), (26, 1)]:
    got, td = run(v, ch)
    worst = max(max(field_errors(got[r], ref[r])) for r in range(3))
    print("variant", v, "chunk", ch, "max err vs march2 fast:", worst, "density rel", abs(td - tdr) / tdr, flush=True)
a, _ = run(20, 32); b, _ = run(22, 5)
print("stream 128/3 vs 64/3 chunk 5 bit-identical:", all(bits_equal(a[r][k], b[r][k]) for r in range(3) for k in range(7)))
