"""Random shapes, chunk lengths and arithmetic forms against Oracle B through the C ABI (one process, one GPU).
Strict: all three slots bit-identical after sweeps and whole steps.  Fast: <= 1e-13 of the field scale per sweep
from the oracle's state (restored before every sweep).  usage: fuzz_shapes.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import small_case, oracle_params, bits_equal, PROPS
from cfd_amd.solver import AdjGraph
from oracle import OracleB

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
edges2 = [1, 2, 3, 15, 16, 17, 62, 63, 64, 65, 125, 126, 127, 128, 239, 240, 241, 242, 253, 254, 255, 256, 257, 480, 481, 495, 496, 497]
edges3 = [1, 2, 11, 12, 13, 14, 15, 16, 17, 23, 24, 25, 28, 29, 31, 32, 33]
bad = 0
for c in range(ncases):
    three = rng.random() < 0.45
    fast = rng.random() < 0.5
    chunk = int(rng.choice([1, 2, 3, 5, 7, 31, 32, 33]))
    if three:
        nx = int(rng.integers(1, 12)); ny = int(rng.choice(edges3)); nz = int(rng.choice(edges3))
        if nz == 1: nz = 2
    else:
        nx = int(rng.integers(1, 40)); ny = int(rng.choice(edges2)); nz = 1
        if nx * ny > 6000: nx = max(1, 6000 // ny)
    os.environ["CFD_CHUNK"] = str(chunk)
    os.environ.pop("CFD_VARIANT", None)
    params, fields, active = small_case(nx, ny, nz, seed=1000 + c, zero_v=bool(rng.random() < 0.3))
    p = dict(params); p["flags"] = 4 if fast else 0
    g = AdjGraph(p, fields, active)
    b = OracleB(oracle_params(params), fields, active, pow_mode=1, nthreads=4)
    what = "%s %s %dx%dx%d chunk %d" % ("3-D" if three else "2-D", "fast" if fast else "strict", nx, ny, nz, chunk)
    ok = True
    try:
        if not fast:
            for st in (True, False, False):
                g.sweep(st); b.sweep(st)
            g.time_filter(); b.time_filter()
            g.step(2); b.step(2)
            for rel in range(3):
                fg, fb = g.read_fields(rel), b.read_fields(rel)
                for k in range(7):
                    if not bits_equal(fg[k], fb[k]):
                        ok = False; print("MISMATCH", what, "slot", rel, PROPS[k], np.max(np.abs(fg[k] - fb[k])))
        else:
            for it in range(3):
                g.step(1); b.step(1)
                fg, fb = g.read_fields(0), b.read_fields(0)
                vs = max(np.max(np.abs(fb[0])), np.max(np.abs(fb[1])), np.max(np.abs(fb[2])))
                for k in (0, 1, 2, 3, 4, 6):
                    sc = vs if k < 3 else np.max(np.abs(fb[k]))
                    e = np.max(np.abs(fg[k] - fb[k])) / sc if sc > 0 else np.max(np.abs(fg[k] - fb[k]))
                    if not (e <= 3e-13):
                        ok = False; print("ERROR", what, "step", it, PROPS[k], e)
                # judge every step on its own: continue from the oracle's state
                for rel in range(3):
                    g.write_fields(b.read_fields(rel), rel)
    except Exception as ex:
        ok = False; print("EXCEPTION", what, type(ex).__name__, ex)
    g.close(); b.close()
    bad += 0 if ok else 1
    print("ok " if ok else "BAD", what, flush=True)
print("fuzz: %d cases, %d bad" % (ncases, bad))
sys.exit(1 if bad else 0)
