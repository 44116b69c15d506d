"""Two 3-D launch shapes must give the same bits: usage k3_check.py variantA variantB (env N = cube edge, 96)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
n = int(os.environ.get("N", 96))
params, fields, active = cases.synthetic_3d(n, n + 5, n - 7, seed=3)
params["dt"] = 1e-7; params["flags"] = 4
out = []
for v in sys.argv[1:3]:
    os.environ["CFD_VARIANT"] = v
    g = AdjGraph(params, fields, active)
    g.step(2)
    out.append([np.array(a) for a in g.read_fields(0)] + [np.array(a) for a in g.read_fields(1)])
    print(v, g.kernel_name(), flush=True)
same = all(np.array_equal(a, b) for a, b in zip(*out))
print("bit-identical" if same else "DIFFERENT")
sys.exit(0 if same else 1)
