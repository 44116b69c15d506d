"""Slab decomposition along x: one AdjGraph per GPU, one process per GPU.

x is the slowest index of the reference's node order, so a rank's slab is a contiguous range
of x-planes and the halo it owes each neighbour is its first / last owned plane.  The device
layout keeps all live fields of a plane contiguous, so that halo is ONE message per neighbour
per sweep, sent straight from / received straight into the time slot (no pack kernel).

Per sweep:  start the exchange of the current slot's ghost planes, launch every plane that needs no
            ghost plane, wait for the exchange, launch the first / last plane (interior compute
            overlaps the transfer).
Per step:   3 such sweeps; the live filter pass needs no halo and is fused into the third sweep.
            Where the library can do the second and third sweep in one pass (2-D, re-associated
            arithmetic), the step is one such sweep, one exchange of TWO planes per side, and that pass.
            (The library's own driver, cfd_multi_* / cfd_amd.multi, also hides both exchanges of such a
            step behind the first sweep's interior planes; it is what bench.py uses at N > 1.)
Reduction:  local tree sum of Density, then one all-reduce of a single double.

Domain ends keep the reference's clamp-to-self rule (no wrap-around): rank 0 has no low
ghost, the last rank no high ghost.

The transport is pluggable so that the orchestration can be exercised without GPUs:
  DistComm   torch.distributed point-to-point (backend "nccl" = RCCL over xGMI on the GPU box,
             "gloo" in the CPU tests)
  LocalComm  several slabs inside one process (virtual ranks), planes copied directly
"""
import os

import numpy as np

from .solver import AdjGraph


def init_rccl(device, **kw):
    """Process group on the nccl (= RCCL) backend bound to `device` (a torch.device).  The group's transfers run at
    the default stream priority: on this runtime a step that mixes stream priorities has been seen to stall once
    hardware queues have been handed on between streams."""
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=device, **kw)


class _DeviceSpan:
    """Exposes a raw device address range to torch through the CUDA array interface."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class GpuBackend:
    """AdjGraph on this process's GPU; halo planes are handed out as torch tensors that alias the
    library's device memory."""

    def __init__(self, params, fields, active):
        import torch
        self._torch = torch
        self.graph = AdjGraph(params, fields, active)
        plane = params["ny"] * params["nz"]
        act = np.ascontiguousarray(active, dtype=np.uint8).reshape(-1)
        self._plane, self._nx = plane, params["nx"]
        two = min(2, params["nx"]) * plane
        self._first_masks, self._last_masks = act[:two].copy(), act[-two:].copy()      # up to two planes, ascending x
        # run on torch's current stream so that torch.distributed's stream ordering covers the kernels
        self.graph.set_stream(torch.cuda.current_stream().cuda_stream)
        self._cache = {}

    def halo_tensors(self, depth=1):
        ptrs, nbytes = self.graph.halo_planes(depth)
        out = []
        for p in ptrs:
            if not p:
                out.append(None)
                continue
            t = self._cache.get((p, nbytes))
            if t is None:
                t = self._torch.as_tensor(_DeviceSpan(p, nbytes), device="cuda")
                self._cache[(p, nbytes)] = t
            out.append(t)
        return out

    def planes(self):
        return self._nx

    def boundary_active(self, depth=1):
        """This slab's masks of its first and last `depth` planes (host arrays, ascending x)."""
        n = depth * self._plane
        return self._first_masks[:n], self._last_masks[-n:]

    def mask_tensor(self, plane=None, depth=1):
        """A device tensor for exchanging mask planes: a copy of `plane`, or an empty one to receive `depth` planes into."""
        torch = self._torch
        if plane is None:
            return torch.empty(depth * self._plane, dtype=torch.uint8, device="cuda")
        return torch.from_numpy(np.ascontiguousarray(plane)).cuda()

    def halo_commit(self, depth=1):
        """Received planes land directly in the time slot; nothing to unpack."""

    def scalar_tensor(self, v):
        return self._torch.tensor([v], dtype=self._torch.float64, device="cuda")

    def __getattr__(self, name):
        return getattr(self.graph, name)


class DistComm:
    """Neighbour exchange and the scalar all-reduce over torch.distributed."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world

    def _neighbour_ops(self, send_lo, send_hi, recv_lo, recv_hi):
        dist = self.dist
        ops = []
        if self.rank > 0:
            ops.append(dist.P2POp(dist.isend, send_lo, self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, recv_lo, self.rank - 1))
        if self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, send_hi, self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, recv_hi, self.rank + 1))
        return ops

    def exchange_masks(self, backend, depth=1):
        """The neighbours' masks of the `depth` planes next to this slab: (lo, hi) host arrays, None at a domain end."""
        if self.world == 1:
            return None, None
        first, last = backend.boundary_active(depth) if depth > 1 else backend.boundary_active()
        empty = (lambda: backend.mask_tensor(None, depth)) if depth > 1 else backend.mask_tensor
        recv_lo = empty() if self.rank > 0 else None
        recv_hi = empty() if self.rank < self.world - 1 else None
        for req in self.dist.batch_isend_irecv(self._neighbour_ops(backend.mask_tensor(first), backend.mask_tensor(last),
                                                                   recv_lo, recv_hi)):
            req.wait()
        host = lambda t: None if t is None else t.cpu().numpy()
        return host(recv_lo), host(recv_hi)

    def agree_pair(self, slab):
        slab.set_ghost_masks()
        return self.all_min(slab.backend, int(slab.backend.pair_available()))

    def agree_one_pass(self, slab):
        """Whether every rank can run a whole step as one pass behind ONE three-plane exchange: masks two planes deep
        where every rank has two planes to show, then the smallest of the ranks' answers."""
        if not self.all_min(slab.backend, int(slab.backend.planes() >= 2)):
            return False
        slab.set_ghost_masks(depth=2)
        return bool(self.all_min(slab.backend, int(slab.backend.one_pass_step_available())))

    def all_min(self, backend, v):
        """Smallest of the ranks' integers."""
        if self.world == 1:
            return int(v)
        t = backend.scalar_tensor(float(v))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def exchange_start(self, backend, depth=1):
        """Posts the sends of this rank's first / last `depth` owned planes and the receives into its ghost
        planes; returns the requests.  With the nccl backend the transfers run on the process
        group's own stream, ordered after the work already queued on the current stream."""
        if self.world == 1:
            return []
        dist = self.dist
        send_lo, send_hi, recv_lo, recv_hi = backend.halo_tensors(depth)
        ops = []
        if self.rank > 0:
            ops.append(dist.P2POp(dist.isend, send_lo, self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, recv_lo, self.rank - 1))
        if self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, send_hi, self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, recv_hi, self.rank + 1))
        return dist.batch_isend_irecv(ops)

    def exchange_wait(self, backend, reqs, depth=1):
        """Orders the current stream (nccl) or the host (gloo) after the transfers."""
        for req in reqs:
            req.wait()
        backend.halo_commit(depth)

    def exchange(self, backend, depth=1):
        self.exchange_wait(backend, self.exchange_start(backend, depth), depth)

    def all_sum(self, backend, v):
        if self.world == 1:
            return v
        t = backend.scalar_tensor(v)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def agree_two_d(self, mine):
        """Whether EVERY rank's W is identically +0 (a collective, before any backend exists): the device form
        without W is a property of the whole grid -- slabs that disagreed would exchange planes of different
        sizes."""
        if self.world == 1:
            return mine
        import torch
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([1.0 if mine else 0.0], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)


class LocalComm:
    """Virtual ranks: all slabs live in this process; `peers` is the list of SlabGraphs in rank
    order.  Exchange copies planes directly; used to test ghost-plane handling on one device."""

    def __init__(self, rank, peers, all_w_zero=None):
        self.rank, self.peers = rank, peers
        self.world = None
        self.all_w_zero = all_w_zero

    def exchange(self, backend, depth=1):
        # pull model: every rank copies what it needs from its neighbours' send planes
        me = self.rank
        _, _, recv_lo, recv_hi = backend.halo_tensors(depth)
        if me > 0:
            recv_lo.copy_(self.peers[me - 1].backend.halo_tensors(depth)[1])
        if me < len(self.peers) - 1:
            recv_hi.copy_(self.peers[me + 1].backend.halo_tensors(depth)[0])
        backend.halo_commit(depth)

    def exchange_start(self, backend, depth=1):
        return []

    def exchange_wait(self, backend, reqs, depth=1):
        self.exchange(backend, depth)

    def exchange_masks(self, backend, depth=1):
        me = self.rank
        get = (lambda b: b.boundary_active(depth)) if depth > 1 else (lambda b: b.boundary_active())
        lo = get(self.peers[me - 1].backend)[1] if me > 0 else None
        hi = get(self.peers[me + 1].backend)[0] if me < len(self.peers) - 1 else None
        return lo, hi

    def agree_pair(self, slab):
        for g in self.peers:
            g.set_ghost_masks()
        return all(g.backend.pair_available() for g in self.peers)

    def agree_one_pass(self, slab):
        if not all(hasattr(g.backend, "planes") and g.backend.planes() >= 2 for g in self.peers):
            return False
        for g in self.peers:
            g.set_ghost_masks(depth=2)
        return all(g.backend.one_pass_step_available() for g in self.peers)

    def all_sum(self, backend, v):
        """Sum over the virtual ranks, in rank order (every slab computes its own part: called on any of them,
        it returns the whole)."""
        acc = 0.0
        for g in self.peers:
            acc += g.backend.total_density()
        return acc

    def agree_two_d(self, mine):
        """Virtual ranks are built one after the other, so the decision is the caller's: pass all_w_zero to
        every LocalComm (default: each slab judges by its own W, as a single slab does)."""
        return mine if self.all_w_zero is None else bool(self.all_w_zero)


class SlabGraph:
    """One rank's slab.  `fields` / `active` hold this rank's owned planes only."""

    def __init__(self, params, fields, active, rank=0, world=1, comm=None, backend_cls=GpuBackend, pair=True):
        """pair=False keeps the three-sweeps-per-step form (one-plane halos only) even where the one-pass
        tail is available; every rank must pass the same value.  The form of a step that runs both exchanges
        on a side stream beside the first sweep's interior planes lives in the library's own driver
        (cfd_multi_*, cfd_amd.multi), on two non-blocking streams per device."""
        p = dict(params)
        p["ghost_lo"] = 1 if rank > 0 else 0
        p["ghost_hi"] = 1 if rank < world - 1 else 0
        self.rank, self.world = rank, world
        self.comm = comm if comm is not None else DistComm(rank, world)
        # nz == 1: the library drops W where it is identically +0 -- decided on the whole grid, not per slab
        if p["nz"] == 1 and world > 1 and hasattr(self.comm, "agree_two_d"):
            mine = not np.ascontiguousarray(fields[2], dtype=np.float64).view(np.uint64).any()
            everyone = self.comm.agree_two_d(mine)      # a collective: every rank calls it, whatever its own W
            if mine and not everyone:
                p["flags"] = int(p.get("flags", 0)) | 2         # CFD_FLAG_FORCE_3D: another slab has W != 0
        self.backend = backend_cls(p, fields, active)
        self.graph = self.backend.graph if hasattr(self.backend, "graph") else self.backend
        self._masks_set = 0         # depth of the neighbours' masks handed to the backend so far
        self._use_pair = None if pair else False
        self._use_one_pass = None if pair else False

    def set_ghost_masks(self, depth=1):
        """Hands the backend the neighbouring slabs' masks of the `depth` planes next to this slab (once per depth; a
        collective with DistComm)."""
        if self._masks_set < depth and self.world > 1:
            if depth > 1:
                lo, hi = self.comm.exchange_masks(self.backend, depth)
                self.backend.set_ghost_active(lo, hi, depth)
            else:
                lo, hi = self.comm.exchange_masks(self.backend)
                self.backend.set_ghost_active(lo, hi)
        self._masks_set = max(self._masks_set, depth)

    def use_one_pass(self):
        """Whether a whole step (three sweeps and the filter) runs as one pass over memory behind ONE three-plane
        exchange.  The ranks decide together, once (a collective with DistComm)."""
        if self._use_one_pass is None:
            self._use_one_pass = self.world > 1 and hasattr(self.backend, "one_pass_step_available") and \
                hasattr(self.comm, "agree_one_pass") and bool(self.comm.agree_one_pass(self))
        return self._use_one_pass

    def use_pair(self):
        """Whether the tail of a step (second and third sweep + filter) runs as one pass over memory with a
        two-plane halo.  Neighbours must exchange the same depth, so the ranks decide together, once
        (a collective with DistComm)."""
        if self._use_pair is None:
            self._use_pair = self.world > 1 and hasattr(self.backend, "pair_available") and \
                bool(self.comm.agree_pair(self))
        return self._use_pair

    # the reference's verbs, slab-aware
    def exchange(self):
        self.comm.exchange(self.backend)

    def _sweep(self, step_time, with_filter):
        if self.world == 1:
            if with_filter:
                self.backend.sweep_and_filter()
            else:
                self.backend.sweep(step_time)
            return
        reqs = self.comm.exchange_start(self.backend)
        self.backend.sweep_begin(step_time, with_filter)     # planes that read no ghost plane
        self.comm.exchange_wait(self.backend, reqs)
        self.backend.sweep_end()                            # first / last plane, slot rotation

    def sweep(self, step_time):
        self._sweep(step_time, False)
        return self

    def time_filter(self):
        self.backend.time_filter()
        return self

    def step(self, nsteps=1):
        if self.world == 1 and hasattr(self.backend, "step"):
            self.backend.step(nsteps)       # undecomposed: the library's own step (second and third sweep in one pass)
            return self
        if self.use_one_pass():
            for _ in range(nsteps):
                self.comm.exchange(self.backend, depth=3)       # three planes per side: the whole step
                self.backend.step_one_pass()
            return self
        pair = self.use_pair()
        for _ in range(nsteps):
            self._sweep(True, False)
            if pair:
                self.comm.exchange(self.backend, depth=2)       # two planes per side: both remaining sweeps
                self.backend.sweep_pair_and_filter()
            else:
                self._sweep(False, False)
                self._sweep(False, True)
        return self

    def total_density(self):
        return self.comm.all_sum(self.backend, self.backend.total_density())

    def read_fields(self, slot_rel=0):
        return self.backend.read_fields(slot_rel)

    def property_slice(self, prp, z_level=None):
        """This rank's rows of the output slice (adjGraphToGrid, src/Main.hs:339-346) of the current slot.
        Vorticity reads the neighbouring slabs' boundary planes, so the ghost planes are exchanged first
        (collective: every rank calls it)."""
        if self.world != 1:
            self.exchange()
        return self.backend.property_slice(prp, z_level)

    def sync(self):
        self.backend.sync()

    def close(self):
        self.backend.close()


def split_slabs(fields, active, nx, plane, world):
    """Cuts global arrays (node order, x slowest) into `world` slabs of consecutive planes.
    Returns a list of (x_begin, x_end, fields, active)."""
    bounds = [nx * r // world for r in range(world + 1)]
    out = []
    for r in range(world):
        a, b = bounds[r] * plane, bounds[r + 1] * plane
        out.append((bounds[r], bounds[r + 1], [np.ascontiguousarray(f[a:b]) for f in fields],
                    np.ascontiguousarray(active[a:b])))
    return out
