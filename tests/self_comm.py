"""A transport with DistComm's interface whose only peer is the rank itself: what the rank sends up lands in
its own low ghost planes and vice versa (a periodic domain of one slab).  It lets SlabGraph.step run its
N > 1 orchestration -- exchanges through RCCL, side stream, split sweeps -- on a single GPU; the result is
not the reference's (whose domain ends clamp), so runs are compared with each other."""
import torch.distributed as dist


class SelfComm:
    def __init__(self):
        self.rank, self.world = 0, 1

    def exchange_masks(self, backend):
        first, last = backend.boundary_active()
        return last, first

    def agree_pair(self, slab):
        slab.set_ghost_masks()
        return bool(slab.backend.pair_available())

    def exchange_start(self, backend, depth=1):
        send_lo, send_hi, recv_lo, recv_hi = backend.halo_tensors(depth)
        # sends and receives to one peer pair up in order: low planes -> high ghosts, high planes -> low ghosts
        return dist.batch_isend_irecv([dist.P2POp(dist.isend, send_lo, 0), dist.P2POp(dist.irecv, recv_hi, 0),
                                       dist.P2POp(dist.isend, send_hi, 0), dist.P2POp(dist.irecv, recv_lo, 0)])

    def exchange_wait(self, backend, reqs, depth=1):
        for r in reqs:
            r.wait()
        backend.halo_commit(depth)

    def exchange(self, backend, depth=1):
        self.exchange_wait(backend, self.exchange_start(backend, depth), depth)

    def all_sum(self, backend, v):
        return v
