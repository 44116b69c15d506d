"""Runs the shipped case for STEPS steps on the GPU and writes the slices of every Property the way the reference's
(disabled) plotting block would: <out>/<Property>[ LOGSCALE]/<step>.png plus CSV.  usage: dump_slices.py OUT [STEPS]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_amd import cases, output
from cfd_amd.solver import AdjGraph

out = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
params, fields, active = cases.reference_case()
g = AdjGraph(params, fields, active)
g.step(steps)
output.dump_step(g, out, steps - 1)
print("wrote", out, "total density", g.total_density())
