"""What the slab path costs on one GPU: a 4096 x 4096 slab with both ghost planes, stepped with whole
sweeps and with split sweeps (interior launch, ghost-plane copy standing in for the exchange, boundary
launches) and with the one-pass tail behind a two-plane copy, against the undecomposed handle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfd_amd import cases
from cfd_amd.slab import GpuBackend
params, fields, active = cases.synthetic_2d(4096, 4096, seed=1)
params["dt"] = 1e-7; params["flags"] = 4
def timed(make, step, n=200, warm=20):
    b = make()
    for _ in range(warm): step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    b.close(); return dt * 1e3
plain = timed(lambda: GpuBackend(params, fields, active), lambda b: b.step(1))
pg = dict(params, ghost_lo=1, ghost_hi=1)
def whole(b):
    for i, st in enumerate((True, False, False)):
        sl, sh, rl, rh = b.halo_tensors(); rl.copy_(sl); rh.copy_(sh)
        (b.sweep_and_filter() if i == 2 else b.sweep(st))
def split(b):
    for i, st in enumerate((True, False, False)):
        sl, sh, rl, rh = b.halo_tensors()
        b.sweep_begin(st, i == 2); rl.copy_(sl); rh.copy_(sh); b.sweep_end()
import numpy as np
ones = np.ones(4096, np.uint8)
def make_pair():
    b = GpuBackend(pg, fields, active); b.set_ghost_active(ones, ones); assert b.pair_available(); return b
def pair(b):          # what SlabGraph.step does at N > 1: split first sweep, two-plane exchange, one-pass tail
    sl, sh, rl, rh = b.halo_tensors()
    b.sweep_begin(True, False); rl.copy_(sl); rh.copy_(sh); b.sweep_end()
    sl, sh, rl, rh = b.halo_tensors(2); rl.copy_(sl); rh.copy_(sh)
    b.sweep_pair_and_filter()
w = timed(lambda: GpuBackend(pg, fields, active), whole)
s = timed(lambda: GpuBackend(pg, fields, active), split)
pr = timed(make_pair, pair)
print("ms/step: undecomposed %.4f   slab, whole sweeps %.4f   slab, split sweeps %.4f   slab, split sweep + one-pass tail %.4f"
      % (plain, w, s, pr))
