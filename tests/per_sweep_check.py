import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import field_errors, oracle_params
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
from oracle import OracleB
params, fields, active = cases.reference_case()
b = OracleB(oracle_params(params), fields, active, pow_mode=1, nthreads=8)
gs = {}
for name, fl in (("fast-march", 4), ("fast-simple", 5), ("strict-march", 0)):
    p = dict(params); p["flags"] = fl
    gs[name] = AdjGraph(p, fields, active)
for sweep in range(33):
    st = sweep % 3 == 0
    b.sweep(st)
    fb = b.read_fields(0)
    msg = []
    for name, g in gs.items():
        g.sweep(st)
        e = field_errors(g.read_fields(0), fb)
        msg.append("%s %.2e" % (name, max(e)))
    print(sweep, " | ".join(msg), flush=True)
