"""Generates the committed golden fixtures.  Run from the repo root: python tests/golden/make_golden.py

The reference program cannot be built or run in this environment (no GHC), so these vectors come
from oracle/oracle_a.py, the literal restatement of the reference's term algebra, and -- for the
45,451-cell shipped case only, which is too large for the Python interpreter -- from oracle_b in
libm-pow mode (bit-identical to oracle_a on every small case).  The two known answers the reference
itself records (src/old test.txt:2-3) are copied into kat.json as data.
"""
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))

from helpers import oracle_params, small_case  # noqa: E402
from oracle import OracleB, oracle_a as A  # noqa: E402


def graph_of(params, fields, active):
    nx, ny, nz = params["nx"], params["ny"], params["nz"]
    return A.build_graph((nx - 1, ny - 1, nz - 1), None, fields, active,
                         cellLength=[params["dx"], params["dy"], params["dz"]])


def dump_case(name, params, fields, active, sweeps, steps):
    """Inputs plus: fields of the current slot after each sweep of `sweeps` (list of step_time flags),
    then (restarting) all three slots + (cmt, cat) + totalDensity after each of `steps` steps."""
    out = {"params": json.dumps(oracle_params(params)), "active": active.astype(np.uint8)}
    for k in range(7):
        out["in_%d" % k] = np.asarray(fields[k], np.float64)
    g = graph_of(params, fields, active)
    for i, st in enumerate(sweeps):
        g = A.applyDiffEq_adj(A.calcSteps_adj, st, g)
        for k, f in enumerate(A.slot_fields(g, 0)):
            out["sweep%d_%d" % (i, k)] = np.array(f)
    out["sweep_flags"] = np.array(sweeps, np.uint8)
    g = graph_of(params, fields, active)
    times, dens = [], []
    for s in range(steps):
        g = A.runSingleStep(g)
        for rel in range(3):
            for k, f in enumerate(A.slot_fields(g, rel)):
                out["step%d_slot%d_%d" % (s, rel, k)] = np.array(f)
        times.append((g.currModTime, g.currAbsTime))
        dens.append(A.totalDensity(g))
    out["step_times"] = np.array(times, np.int64)
    out["step_density"] = np.array(dens, np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


CASES = ("case_12x9x1", "case_7x6x5", "case_3x3x3_zero_vw", "case_8x7x1_zero_v")


def dump_slices(name):
    """Output slices (adjGraphToGrid, src/Main.hs:339-346) of every Property at every z-level, evaluated by
    oracle_a on the state the case fixture records after its last step: the fields of slot currModTime there
    are the inputs (key step<last>_slot0_<k> of <name>.npz), the slices are <name>_slices.npz[z<level>_p<property>]."""
    d = np.load(os.path.join(HERE, name + ".npz"))
    params = json.loads(str(d["params"]))
    last = len(d["step_density"]) - 1
    fields = [d["step%d_slot0_%d" % (last, k)] for k in range(7)]
    nx, ny, nz = params["nx"], params["ny"], params["nz"]
    g = A.build_graph((nx - 1, ny - 1, nz - 1), None, fields, d["active"],
                      cellLength=[params["dx"], params["dy"], params["dz"]])
    out = {"last_step": np.int64(last)}
    for zl in range(nz):
        for prp in range(10):
            out["z%d_p%d" % (zl, prp)] = np.array(A.adjGraphToGrid(g, prp, ny, zl), np.float64)
    np.savez_compressed(os.path.join(HERE, name + "_slices.npz"), **out)
    print("wrote", name + "_slices")


def main():
    if "--slices-only" in sys.argv:         # the case fixtures stay as committed
        for name in CASES:
            dump_slices(name)
        return
    kat = {
        "solveUnknown_testEquation": 0.5866666666666668,           # src/old test.txt:2
        # src/old test.txt:3, printed in reverse term order there
        "distributeMultiply_testTerms_2": [["Unknown", 4.8], ["Constant", 2.4], ["Constant", 6.224],
                                           ["Unknown", -0.42], ["Constant", 4.0], ["Constant", 4.0],
                                           ["Unknown", 0.66666]],
        "sutherlandLambda": A.sutherlandLambda,
        "totalDensity_step0_shipped_case": 13635.299999990304,
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    p, fl, ac = small_case(12, 9, 1, seed=101)
    dump_case("case_12x9x1", p, fl, ac, [True, False, False], 3)
    p, fl, ac = small_case(7, 6, 5, seed=102)
    dump_case("case_7x6x5", p, fl, ac, [True, False, False], 2)
    # exact-zero V and W everywhere (NaN -> 1 coefficient rule), 3x3x3 neighbourhood
    p, fl, ac = small_case(3, 3, 3, seed=103, zero_v=True)
    fl[2][:] = 0.0
    dump_case("case_3x3x3_zero_vw", p, fl, ac, [True, False], 1)
    # 2-D with V exactly zero
    p, fl, ac = small_case(8, 7, 1, seed=104, zero_v=True)
    dump_case("case_8x7x1_zero_v", p, fl, ac, [True, False, False], 2)

    for name in CASES:
        dump_slices(name)

    # first five stdout blocks of the shipped case (src/Main.hs:296-301)
    from cfd_amd import cases
    from cfd_amd.solver import _hs_show_double
    params, fields, active = cases.reference_case()
    b = OracleB(oracle_params(params), fields, active, pow_mode=0, nthreads=8)
    buf = io.StringIO()
    for step in range(5):
        b.step(1)
        cmt, cat = b.time()
        buf.write("%d\nstep: %d\ntimeLevel: %d\ntimeLevelAbsolute: %d\ntotalDensity: %s\n \n"
                  % (b.n, step, cmt, cat, _hs_show_double(b.total_density())))
    with open(os.path.join(HERE, "shipped_case_stdout_5steps.txt"), "w") as f:
        f.write(buf.getvalue())
    print(buf.getvalue())


if __name__ == "__main__":
    main()
