"""A middle rank's step through the library's own slab driver and RCCL, rehearsed on ONE GPU: cfd_multi_create_rank with
world 1 and CFD_MULTI_LOOPBACK=1 makes the slab its own lower and upper neighbour -- every step it exchanges its ghost
planes with itself by grouped ncclSend / ncclRecv on the side stream, exactly the calls a rank between two others makes.
Printed beside the undecomposed step, alternating, with the duration of the exchanges.  (A periodic domain: for timing.)
usage: loopback_rank_step.py [steps]     env NZ (1: 4096^2; else NZ^3), FLAGS (4 fast, 0 strict)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfd_amd import cases
from cfd_amd.solver import AdjGraph
from cfd_amd.multi import MultiGraph

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nz = int(os.environ.get("NZ", "1"))
if nz > 1:
    params, fields, active = cases.synthetic_3d(nz, nz, nz, seed=3)
else:
    params, fields, active = cases.synthetic_2d(4096, 4096, seed=1)
params = dict(params, dt=1e-7, flags=int(os.environ.get("FLAGS", "4")), device=0)
und = AdjGraph(params, fields, active)
os.environ["CFD_MULTI_LOOPBACK"] = "1"
lb = MultiGraph.for_rank(params, fields, active, 0, 1, None)
print("loopback slab: (local slabs, step form, overlapped) =", lb.info(), flush=True)


def timed(fn, sync):
    sync()
    t0 = time.perf_counter()
    fn(steps)
    t1 = time.perf_counter()
    sync()
    return (time.perf_counter() - t0) / steps * 1e3, (t1 - t0) / steps * 1e3


warm = 300 if nz == 1 else 20
und.step(warm); lb.step(warm); und.sync(); lb.sync()
for r in range(3):
    print("undecomposed                       %.4f ms/step (host issue %.4f)" % timed(und.step, und.sync))
    lb.set_exchange_timing(True)
    t = timed(lb.step, lb.sync)
    ms, n = lb.exchange_timing(reset=True)
    lb.set_exchange_timing(False)
    print("rank between two (RCCL, loopback)  %.4f ms/step (host issue %.4f); %d exchanges per step, %.1f us each" % (
        t + (n // steps, ms / max(n, 1) * 1e3)), flush=True)
