"""Prints, for one fast sweep from the oracle's state, the per-field error in three metrics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
from helpers import small_case, oracle_params, field_errors, elem_rel_errors, PROPS
from oracle import OracleB
for name, mk in (("small", lambda: small_case(40, 33, 1, 13)), ("synthetic", lambda: cases.synthetic_2d(300, 280, 1)),
                 ("cavity", lambda: cases.cavity_2d(64, 64)), ("small3d", lambda: small_case(21, 18, 19, 5)),
                 ("synthetic_dt1e-7", lambda: (lambda p, f, a: (dict(p, dt=1e-7), f, a))(*cases.synthetic_2d(300, 280, 1)))):
    params, fields, active = mk()
    b = OracleB(oracle_params(params), fields, active, pow_mode=1, nthreads=4)
    b0 = OracleB(oracle_params(params), fields, active, pow_mode=0, nthreads=4)
    g = AdjGraph(dict(params, flags=4), fields, active)
    g.sweep(True); b.sweep(True); b0.sweep(True)
    got, want, w0 = g.read_fields(0), b.read_fields(0), b0.read_fields(0)
    own = [float(np.max(np.abs(got[k] - want[k])) / max(np.max(np.abs(want[k])), 1e-300)) for k in range(7)]
    print(name, "dt", params["dt"])
    print("  vector-scale ", ["%.1e" % e for e in field_errors(got, want)])
    print("  own-scale    ", ["%.1e" % e for e in own])
    for fl in (1e-3, 1e-2, 1e-1):
        print("  elem floor %g" % fl, ["%.1e" % e for e in elem_rel_errors(got, want, fl)])
    print("  oracle pow-vs-sqrt elem 1e-3", ["%.1e" % e for e in elem_rel_errors(w0, want, 1e-3)])
