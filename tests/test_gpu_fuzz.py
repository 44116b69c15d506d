"""Seeded random shapes, chunk lengths and arithmetic forms against Oracle B through the C ABI.

The hand-picked shapes of test_gpu_parity.py straddle the tile edges the author thought of; these draw the extents
from every value next to a tile, wave or cache-line boundary of the launch shapes (2-D: 62/63/64, 126/128, 240/241,
248/249, 254/256, two tiles 480/496; 3-D: 12/14/16 and their multiples) and combine them with chunk lengths of 1 to 7, the
default 32 and primes, strict and re-associated arithmetic, with and without an identically zero V."""
import numpy as np
import pytest

from helpers import PROPS, bits_equal, oracle_params, small_case

pytestmark = pytest.mark.gpu

EDGES_2D = [1, 2, 3, 7, 8, 9, 15, 16, 17, 62, 63, 64, 65, 125, 126, 127, 128, 239, 240, 241, 242, 247, 248, 249, 253, 254, 255,
            256, 257, 480, 481, 495, 496, 497]
EDGES_3D = [1, 2, 11, 12, 13, 14, 15, 16, 17, 23, 24, 25, 28, 29, 31, 32, 33]


@pytest.mark.parametrize("seed", [7, 11, 2026])
def test_random_shapes_against_the_oracle(seed, monkeypatch):
    from cfd_amd.solver import AdjGraph
    from oracle import OracleB
    rng = np.random.default_rng(seed)
    monkeypatch.delenv("CFD_VARIANT", raising=False)
    for c in range(40):
        three = rng.random() < 0.45
        fast = rng.random() < 0.5
        chunk = int(rng.choice([1, 2, 3, 5, 7, 31, 32, 33]))
        if three:
            nx, ny, nz = int(rng.integers(1, 12)), int(rng.choice(EDGES_3D)), max(2, int(rng.choice(EDGES_3D)))
        else:
            nx, ny, nz = int(rng.integers(1, 40)), int(rng.choice(EDGES_2D)), 1
            nx = min(nx, max(1, 6000 // ny))
        monkeypatch.setenv("CFD_CHUNK", str(chunk))
        # planes per block of the one-launch step (2-D, re-associated arithmetic): chosen by the library, or forced short
        trio_chunk = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 31]))
        if trio_chunk:
            monkeypatch.setenv("CFD_TRIO_CHUNK", str(trio_chunk))
        else:
            monkeypatch.delenv("CFD_TRIO_CHUNK", raising=False)
        params, fields, active = small_case(nx, ny, nz, seed=1000 * seed + c, zero_v=bool(rng.random() < 0.3))
        p = dict(params)
        p["flags"] = 4 if fast else 0
        what = "%s %s %dx%dx%d chunk %d / %d" % ("3-D" if three else "2-D", "fast" if fast else "strict", nx, ny, nz, chunk,
                                                 trio_chunk)
        g = AdjGraph(p, fields, active)
        b = OracleB(oracle_params(params), fields, active, pow_mode=1, nthreads=4)
        if not fast:
            # bit for bit: three sweeps, the filter on its own, two whole steps
            for st in (True, False, False):
                g.sweep(st)
                b.sweep(st)
            g.time_filter()
            b.time_filter()
            g.step(2)
            b.step(2)
            for rel in range(3):
                fg, fb = g.read_fields(rel), b.read_fields(rel)
                for k in range(7):
                    assert bits_equal(fg[k], fb[k]), "%s: slot -%d %s" % (what, rel, PROPS[k])
        else:
            # every step judged on its own (the state is put back from the oracle): 3e-13 of the field scale
            for it in range(3):
                g.step(1)
                b.step(1)
                fg, fb = g.read_fields(0), b.read_fields(0)
                vs = max(np.max(np.abs(fb[k])) for k in range(3))
                for k in (0, 1, 2, 3, 4, 6):
                    sc = vs if k < 3 else np.max(np.abs(fb[k]))
                    d = np.max(np.abs(fg[k] - fb[k]))
                    assert d <= 3e-13 * sc or d == 0.0, "%s: step %d %s %.3e" % (what, it, PROPS[k], d / sc if sc else d)
                for rel in range(3):
                    g.write_fields(b.read_fields(rel), rel)
        g.close()
        b.close()
