"""Random shapes, slab counts, arithmetic forms and chunk plans: a decomposed run (library slab driver, copy transport, all slabs
on this GPU) against the undecomposed one, three steps, all slots bit for bit.  usage: slab_fuzz.py [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import small_case, bits_equal
from cfd_amd.solver import AdjGraph
from cfd_amd.multi import MultiGraph
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for c in range(60):
    n = int(rng.integers(2, 6))
    two_d = rng.random() < 0.7
    if two_d:
        nx, ny, nz = int(rng.integers(n * 1, n * 30)), int(rng.choice([7, 33, 240, 248, 249, 300, 500, 1000])), 1
    else:
        nx, ny, nz = int(rng.integers(n, n * 6)), int(rng.choice([5, 12, 17, 30])), int(rng.choice([4, 16, 19]))
    flags = int(rng.choice([4, 4, 0]))
    for k, v in (("CFD_TRIO_CHUNK", rng.choice(["", "1", "3", "8"])), ("CFD_TRIO_EDGE_MARGIN", rng.choice(["", "1", "5"]))):
        if v: os.environ[k] = str(v)
        else: os.environ.pop(k, None)
    params, fields, active = small_case(nx, ny, nz, 1000 + c)
    p = dict(params, flags=flags)
    g = AdjGraph(p, fields, active)
    m = MultiGraph(p, fields, active, n, devices=[0] * n)
    ok = True
    for step in range(3):
        g.step(1); m.step(1)
        for rel in range(3):
            fm, fg = m.read_fields(rel), g.read_fields(rel)
            ok = ok and all(bits_equal(fm[k], fg[k]) for k in range(7))
    print(c, n, (nx, ny, nz), flags, m.info(), "ok" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
    m.close(); g.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
