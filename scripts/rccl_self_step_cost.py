"""What the transport adds to a slab step, measured on one GPU: a 4096 x 4096 slab with ghost planes on both
sides stepped by SlabGraph.step's N > 1 forms, with the exchanges done by RCCL (nccl backend, world size 1,
the rank as its own neighbour on both sides: tests/self_comm.py), against the undecomposed step.  A transfer
to self crosses no link, so the difference is the launch, stream-ordering and host cost of the send/recv pairs
and how much of it each form hides."""
import os, socket, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from cfd_amd import cases
from cfd_amd.slab import SlabGraph
from self_comm import SelfComm

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
from cfd_amd.slab import init_rccl
init_rccl(torch.device("cuda", 0), high_priority=os.environ.get("HIPRI", "1") == "1", rank=0, world_size=1)
params, fields, active = cases.synthetic_2d(int(os.environ.get("NX", "4096")), 4096, seed=1)
params["dt"] = 1e-7; params["flags"] = 4


def timed(make, n=200, warm=20):
    g = make()
    g.step(warm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.step(n)
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    g.close(); return dt * 1e3, t_issue * 1e3


forms = (("undecomposed", lambda: SlabGraph(params, fields, active)),
         ("both exchanges beside the interior", lambda: SlabGraph(params, fields, active, rank=1, world=3, comm=SelfComm())),
         ("two-plane exchange behind the first sweep", lambda: SlabGraph(params, fields, active, rank=1, world=3, comm=SelfComm(), overlap=False)),
         ("three sweeps, one-plane exchanges", lambda: SlabGraph(params, fields, active, rank=1, world=3, comm=SelfComm(), pair=False)))
# FORMS: which of the forms to run (indices, e.g. "1" or "0,2"); STEPS / REPS: timed steps and repetitions
pick = [int(i) for i in os.environ.get("FORMS", "0,1,2,3").split(",")]
nsteps, reps = int(os.environ.get("STEPS", "200")), int(os.environ.get("REPS", "2"))
for rep in range(reps):
    for name, make in [forms[i] for i in pick]:
        ms, issue = timed(make, nsteps, max(2, nsteps // 10))
        print("%-44s %.4f ms/step (host issue %.4f ms/step)" % (name, ms, issue), flush=True)
dist.barrier(); dist.destroy_process_group()
