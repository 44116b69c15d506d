"""40 rounds of creating, stepping and destroying an undecomposed graph and a loopback slab (communicator, streams, events): what
repeated pre-flights would leave behind.  Prints the free device memory every ten rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
from cfd_amd.multi import MultiGraph
import torch
os.environ["CFD_MULTI_LOOPBACK"] = "1"
p, f, a = cases.synthetic_2d(96, 528, seed=3)
p = dict(p, dt=1e-7, flags=4, device=0)
t0 = time.time()
for i in range(40):
    g = AdjGraph(p, f, a)
    m = MultiGraph.for_rank(p, f, a, 0, 1, None)
    m.step(5); g.step(5)
    d = m.total_density()
    m.close(); g.close()
    if i % 10 == 9:
        free, total = torch.cuda.mem_get_info()
        print(i + 1, "cycles", "%.1f s" % (time.time() - t0), "free GPU memory %.2f GB" % (free / 1e9), "density", d, flush=True)
print("done")
