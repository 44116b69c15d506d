"""The multi-rank path on CPU: world_size 2 and 3 over gloo, one process per rank, each rank owning
a slab of x-planes.  Compute is the CPU oracle (tests/slab_double.py); what is under test is
cfd_amd.slab -- ghost flags, which plane goes to which neighbour, exchange-before-every-sweep, the
scalar all-reduce.  Bar: every rank's owned planes equal the single-domain run BIT-FOR-BIT after
two steps, and so do its rows of the ten output slices; the reduced density agrees to 1e-12
relative (summation order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, shape, pair, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from helpers import small_case
    from cfd_amd.slab import DistComm, SlabGraph, split_slabs
    from slab_double import OracleSlabBackend
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        nx, ny, nz = shape
        params, fields, active = small_case(nx, ny, nz, seed=21)
        x0, x1, f, a = split_slabs(fields, active, nx, ny * nz, world)[rank]
        p = dict(params)
        p["nx"] = x1 - x0
        g = SlabGraph(p, f, a, rank=rank, world=world, comm=DistComm(rank, world), backend_cls=OracleSlabBackend,
                      pair=pair)
        g.step(2)
        td = g.total_density()
        slices = [g.property_slice(k) for k in range(10)]      # collective: exchanges the ghost planes
        q.put((rank, x0, x1, [g.read_fields(r) for r in range(3)], td, slices, g.use_pair()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


# pair: the tail of a step as one pass behind a two-plane exchange (True) or as two sweeps with a one-plane
# exchange each (False); used: what the ranks must have agreed on (a one-plane slab cannot send two planes,
# and then no rank may use the two-plane form)
@pytest.mark.parametrize("world,shape,pair,used", [
    (2, (12, 9, 1), True, True), (3, (11, 6, 1), True, True), (2, (8, 5, 4), True, True),
    (2, (12, 9, 1), False, False), (3, (11, 6, 1), False, False),
    (3, (4, 7, 1), True, False), (2, (4, 6, 1), True, True)])
def test_slabs_match_single_domain(world, shape, pair, used):
    sys.path.insert(0, HERE)
    from helpers import bits_equal, oracle_params, small_case
    from oracle import OracleB
    nx, ny, nz = shape
    params, fields, active = small_case(nx, ny, nz, seed=21)
    ref = OracleB(oracle_params(params), fields, active, pow_mode=1)
    ref.step(2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, shape, pair, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    plane = ny * nz
    for rank, x0, x1, slots, td, slices, paired in results:
        assert paired == used, rank
        for k in range(10):     # output slices (Vorticity reaches into the neighbouring slabs)
            assert bits_equal(slices[k], ref.property_slice(k, (nz - 1) // 2)[x0:x1]), (rank, k)
        for rel in range(3):
            want = ref.read_fields(rel)
            for k in range(7):
                assert bits_equal(slots[rel][k], want[k][x0 * plane:x1 * plane]), (rank, rel, k)
        assert abs(td - ref.total_density()) <= 1e-12 * abs(ref.total_density())


def _rank_w(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from helpers import small_case
    from cfd_amd.slab import DistComm, SlabGraph, split_slabs
    from slab_double import OracleSlabBackend
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        nx, ny = 12, 9
        params, fields, active = small_case(nx, ny, 1, seed=33)
        fields = [np.array(f, dtype=np.float64) for f in fields]
        fields[2][-ny * 2:] = 0.03          # W != 0 in the last two planes only: the last rank's slab
        x0, x1, f, a = split_slabs(fields, active, nx, ny, world)[rank]
        p = dict(params)
        p["nx"] = x1 - x0
        g = SlabGraph(p, f, a, rank=rank, world=world, comm=DistComm(rank, world), backend_cls=OracleSlabBackend)
        g.step(2)
        q.put((rank, x0, x1, g.backend.flags, g.read_fields(0)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_w_in_one_slab_only_forces_the_general_form_on_every_rank():
    """nz == 1 and W identically zero on rank 0 but not on rank 1: the ranks agree (all-reduce) before they create
    their handles, and the rank with the zero W asks for CFD_FLAG_FORCE_3D, so that both hold six fields and
    exchange planes of one size; W is advected into rank 0's slab as in the undecomposed run."""
    sys.path.insert(0, HERE)
    from helpers import bits_equal, oracle_params, small_case
    from oracle import OracleB
    nx, ny = 12, 9
    params, fields, active = small_case(nx, ny, 1, seed=33)
    fields = [np.array(f, dtype=np.float64) for f in fields]
    fields[2][-ny * 2:] = 0.03
    ref = OracleB(oracle_params(params), fields, active, pow_mode=1)
    ref.step(2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_w, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = ref.read_fields(0)
    for rank, x0, x1, flags, got in results:
        assert bool(flags & 2) == (rank == 0), (rank, flags)        # rank 1 keeps W by itself
        for k in range(7):
            assert bits_equal(got[k], want[k][x0 * ny:x1 * ny]), (rank, k)


def _preflight_rank(rank, world, port, spoil, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from helpers import oracle_params
    from oracle import OracleB
    from cfd_amd import preflight
    from cfd_amd.slab import DistComm, SlabGraph
    from slab_double import OracleSlabBackend
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)

    class Whole:
        """The undecomposed run on the CPU oracle, with the verbs the pre-flight uses."""

        def __init__(self, params, fields, active):
            self.o = OracleB(oracle_params(params), fields, active, pow_mode=1)
        step = lambda self, n: self.o.step(n)
        read_fields = lambda self, rel: self.o.read_fields(rel)
        total_density = lambda self: self.o.total_density()
        close = lambda self: self.o.close()

    class Slab:
        def __init__(self, params, fields, active):
            self.g = SlabGraph(params, fields, active, rank=rank, world=world, comm=DistComm(rank, world),
                               backend_cls=OracleSlabBackend)
        step = lambda self, n: self.g.step(n)
        total_density = lambda self: self.g.total_density()
        close = lambda self: None

        def read_fields(self, rel):
            out = [np.array(f) for f in self.g.read_fields(rel)]
            if spoil and rank == world - 1 and rel == 1:
                out[3][5] = np.nextafter(out[3][5], 1e9)        # one Density value of the last rank off by one ulp
            return out

    try:
        rep = preflight.run(rank, world, 0, 1, lambda name, p, f, m, x0, x1: Slab(p, f, m), steps=2, make_ref=Whole)
        q.put((rank, rep))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("spoil", [False, True])
def test_preflight_protocol_over_gloo(spoil):
    """The pre-flight a multi-GPU bench runs before it times anything, with two ranks over gloo and the CPU oracle in
    place of the GPU on both sides of the comparison: every rank compares its planes of the decomposed run with the
    undecomposed one bit for bit; a single value that is off by one ulp on one rank is named by that rank only."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_preflight_rank, args=(r, world, port, spoil, q)) for r in range(world)]
    for p in procs:
        p.start()
    reports = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        rep = reports[rank]
        assert len(rep["cases"]) == 2 and rep["steps"] == 2
        if spoil and rank == world - 1:
            assert not rep["ok"] and "rank 1 of 2" in rep["detail"] and "slot -1" in rep["detail"] and "Density" in rep["detail"]
        else:
            assert rep["ok"] and rep["detail"] is None, rep
