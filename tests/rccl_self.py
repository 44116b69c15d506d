"""RCCL smoke on one GPU: world size 1, halo planes of a slab handle sent to self through
torch.distributed (nccl backend), checking tensor aliasing of the library's memory and stream ordering."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, torch.distributed as dist
from cfd_amd.slab import GpuBackend, DistComm
from helpers import small_case
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
from cfd_amd.slab import init_rccl
init_rccl(torch.device("cuda", 0), rank=0, world_size=1)       # as bench.py sets it up
params, fields, active = small_case(12, 40, 1, 5)
p = dict(params); p["ghost_lo"] = 1; p["ghost_hi"] = 1; p["flags"] = 4
b = GpuBackend(p, fields, active)
send_lo, send_hi, recv_lo, recv_hi = b.halo_tensors()
ops = [dist.P2POp(dist.isend, send_hi, 0), dist.P2POp(dist.irecv, recv_lo, 0)]
for r in dist.batch_isend_irecv(ops): r.wait()
torch.cuda.synchronize()
print("after first exchange:", bool(torch.equal(recv_lo.view(torch.int64), send_hi.view(torch.int64))), float(recv_lo.abs().sum()), float(send_hi.abs().sum()), recv_lo.data_ptr(), send_hi.data_ptr(), recv_lo.numel())
ops = [dist.P2POp(dist.isend, send_lo, 0), dist.P2POp(dist.irecv, recv_hi, 0)]
reqs = dist.batch_isend_irecv(ops)
b.sweep_begin(False)
for r in reqs: r.wait()
b.sweep_end()
torch.cuda.synchronize()
t = torch.tensor([b.total_density()], dtype=torch.float64, device="cuda"); dist.all_reduce(t)
print("second exchange ok:", bool(torch.equal(recv_hi.view(torch.int64), send_lo.view(torch.int64))), "all_reduce", float(t.item()),
      "finite after sweep:", bool(torch.isfinite(b.halo_tensors()[0]).all()))
# two planes per side (the halo of the one-pass tail) and a mask plane (uint8), again to self
import numpy as np
b.set_ghost_active(np.ones(40, np.uint8), np.ones(40, np.uint8))
s2_lo, s2_hi, r2_lo, r2_hi = b.halo_tensors(2)
m_send, m_recv = b.mask_tensor(b.boundary_active()[0]), b.mask_tensor()
ops = [dist.P2POp(dist.isend, s2_hi, 0), dist.P2POp(dist.irecv, r2_lo, 0)]
for r in dist.batch_isend_irecv(ops): r.wait()
ops = [dist.P2POp(dist.isend, s2_lo, 0), dist.P2POp(dist.irecv, r2_hi, 0)]
for r in dist.batch_isend_irecv(ops): r.wait()
ops = [dist.P2POp(dist.isend, m_send, 0), dist.P2POp(dist.irecv, m_recv, 0)]
for r in dist.batch_isend_irecv(ops): r.wait()
b.sweep_pair_and_filter()
torch.cuda.synchronize()
print("two-plane exchange ok:", bool(torch.equal(r2_lo.view(torch.int64), s2_hi.view(torch.int64))) and
      bool(torch.equal(r2_hi.view(torch.int64), s2_lo.view(torch.int64))), r2_lo.numel(), "mask ok:", bool(torch.equal(m_send, m_recv)),
      "pair available:", b.pair_available(), "finite after the one-pass tail:", bool(torch.isfinite(b.halo_tensors(2)[0]).all()))
# SlabGraph.step's N > 1 forms with the rank as its own neighbour on both sides (tests/self_comm.py): the one-pass
# tail behind a two-plane exchange against three sweeps with one-plane exchanges
from cfd_amd.slab import SlabGraph
from self_comm import SelfComm
from helpers import bits_equal
params, fields, active = small_case(70, 300, 1, 9)
p = dict(params); p["flags"] = 4
runs = []
for kw in (dict(), dict(pair=False)):
    g = SlabGraph(p, fields, active, rank=1, world=3, comm=SelfComm(), **kw)
    g.step(4)
    torch.cuda.synchronize()
    runs.append(([g.read_fields(r) for r in range(3)], g.use_pair()))
    g.close()
same = all(bits_equal(runs[0][0][r][k], runs[1][0][r][k]) for r in range(3) for k in range(7))
print("slab step forms agree:", same, "forms:", [u for _, u in runs], "finite:", bool(np.isfinite(runs[0][0][0][3]).all()))
dist.barrier(); dist.destroy_process_group()
