"""The slab decomposition behind the C ABI (cfd_multi_*): several slabs driven by the library itself, halo
planes moved by device copies (any number of slabs on the one GPU of the test box) -- bit-identical to the
undecomposed run in all three slots; the density sum to 1e-12; RCCL itself through a one-rank communicator."""
import numpy as np
import pytest

from helpers import PROPS, bits_equal, small_case

pytestmark = pytest.mark.gpu
FAST = 4


def _single(params, fields, active, flags):
    from cfd_amd.solver import AdjGraph
    return AdjGraph(dict(params, flags=flags), fields, active)


def _multi(params, fields, active, flags, n, **kw):
    from cfd_amd.multi import MultiGraph
    return MultiGraph(dict(params, flags=flags), fields, active, n, devices=[0] * n, **kw)


def _assert_same(m, g, what):
    assert (m.currModTime, m.currAbsTime) == (g.currModTime, g.currAbsTime)
    for rel in range(3):
        fm, fg = m.read_fields(rel), g.read_fields(rel)
        for k in range(7):
            assert bits_equal(fm[k], fg[k]), (what, rel, PROPS[k], float(np.max(np.abs(fm[k] - fg[k]))))
    a, b = m.total_density(), g.total_density()
    assert abs(a - b) <= 1e-12 * abs(b)


@pytest.mark.parametrize("n,shape,flags", [(2, (12, 33, 1), FAST), (3, (15, 300, 1), FAST), (5, (71, 500, 1), FAST),
                                           (4, (9, 20, 1), FAST), (3, (7, 50, 1), FAST), (2, (40, 33, 1), 0),
                                           (3, (11, 6, 5), FAST), (4, (21, 18, 19), FAST), (2, (9, 8, 7), 0)])
@pytest.mark.parametrize("overlap", ["1", "0"])
@pytest.mark.parametrize("trio,trio_chunk", [("1", None), ("1", 2), ("1", 5), ("0", None)])
def test_virtual_slabs_behind_the_c_abi_match_the_undecomposed_run(n, shape, flags, overlap, trio, trio_chunk, monkeypatch):
    """trio: the slabs run the whole step as one pass (one three-plane exchange per step; chunk lengths that make
    every block, some blocks or no block of a slab touch a ghost plane) or the plain sweep + the one-pass tail
    (two exchanges per step); where a slab has too few planes, or in 3-D / strict, three sweeps with an exchange each."""
    if trio_chunk is not None and not (flags == FAST and shape[2] == 1):
        pytest.skip("the chunk length of the one-pass step only matters where that step runs")
    monkeypatch.setenv("CFD_MULTI_OVERLAP", overlap)
    monkeypatch.setenv("CFD_MULTI_TRIO", trio)
    if trio_chunk is not None:
        monkeypatch.setenv("CFD_TRIO_CHUNK", str(trio_chunk))
    params, fields, active = small_case(shape[0], shape[1], shape[2], 61)
    # inactive cells on the planes next to a slab boundary: the once- and twice-swept ghost values depend on their masks
    act = np.array(active).reshape(shape)
    act[shape[0] // n, ::3, :] = 0
    act[max(shape[0] // n - 1, 0), 1::4, :] = 0
    act[max(shape[0] // n - 2, 0), 2::5, :] = 0
    act[min(shape[0] // n + 1, shape[0] - 1), 3::4, :] = 0
    active = act.reshape(-1)
    g = _single(params, fields, active, flags)
    m = _multi(params, fields, active, flags, n)
    local, one_pass, overlapped = m.info()
    assert local == n
    fewest = min((shape[0] * (i + 1)) // n - (shape[0] * i) // n for i in range(n))
    assert overlapped == (overlap == "1")
    if flags == FAST and shape[2] == 1 and fewest >= 2:
        assert one_pass == (2 if (trio == "1" and fewest >= 3) else 1)
    else:
        assert one_pass == 0
    for step in range(3):
        g.step(1)
        m.step(1)
        _assert_same(m, g, "step %d" % step)
    g.step(2)
    m.step(2)
    _assert_same(m, g, "two more steps")
    if one_pass == 2:       # and the launches really were the one-pass step
        sl = m.slab(0)
        sl.set_timing(True)
        m.step(1); m.sync()
        assert sl.kernel_timing(3)[1] >= 1 and sl.kernel_timing(0)[1] == 0 and sl.kernel_timing(2)[1] == 0


@pytest.mark.parametrize("n,shape", [(2, (64, 70)), (3, (50, 300)), (3, (9, 40))])
@pytest.mark.parametrize("flags", [FAST, 0])
@pytest.mark.parametrize("trio", ["1", "0"])
def test_cavity_cut_into_slabs_matches_the_undecomposed_run(n, shape, flags, trio, monkeypatch):
    """The lid-driven-cavity configuration (U = V = 0 exactly in the interior, so the unknown's coefficient takes
    its other branch in every cell; an inactive ring of walls) through the library's slab driver: ghost planes two
    and three deep, the neighbours' masks, against the undecomposed run, bit for bit, strict and re-associated."""
    from cfd_amd import cases
    monkeypatch.setenv("CFD_MULTI_TRIO", trio)
    params, fields, active = cases.cavity_2d(shape[0], shape[1])
    g = _single(params, fields, active, flags)
    m = _multi(params, fields, active, flags, n)
    for step in range(4):
        g.step(1)
        m.step(1)
        _assert_same(m, g, "cavity step %d" % step)


def test_one_slab_is_the_plain_library():
    params, fields, active = small_case(30, 40, 1, 62)
    g = _single(params, fields, active, FAST)
    m = _multi(params, fields, active, FAST, 1)
    g.step(3); m.step(3)
    _assert_same(m, g, "one slab")


def test_w_in_one_slab_only_keeps_w_everywhere():
    """nz == 1 with W != 0 in the last slab only: whether the device holds W is decided on the whole grid, so the
    first slab (whose own W is identically zero) keeps W too and the planes the slabs exchange have one size."""
    params, fields, active = small_case(16, 12, 1, 63)
    fields = [np.array(f, dtype=np.float64) for f in fields]
    fields[2][-12 * 3:] = 0.05 * np.cos(np.arange(36))
    g = _single(params, fields, active, FAST)
    assert g.live_fields == 6
    m = _multi(params, fields, active, FAST, 2)
    assert m.slab(0).live_fields == 6 and m.slab(1).live_fields == 6
    g.step(3); m.step(3)
    _assert_same(m, g, "W in one slab")
    assert np.any(m.read_fields(0)[2][:12 * 8] != 0.0)         # W has been advected into the first slab


def test_rccl_through_the_library_on_one_gpu():
    """A one-rank RCCL communicator built by the library: grouped send + receive to itself and an all-reduce; and
    the RCCL transport with a single slab (no neighbour) steps like the plain library."""
    from cfd_amd.multi import MultiGraph, comm_unique_id, rccl_selftest
    rccl_selftest(0, 1 << 20)
    assert len(comm_unique_id()) == 128
    params, fields, active = small_case(20, 30, 1, 64)
    g = _single(params, fields, active, FAST)
    m = MultiGraph(dict(params, flags=FAST), fields, active, 1, devices=[0], transport="rccl")
    r = MultiGraph.for_rank(dict(params, flags=FAST, device=0), fields, active, 0, 1, None)
    g.step(2); m.step(2); r.step(2)
    _assert_same(m, g, "rccl, one slab")
    _assert_same(r, g, "rank form, world 1")


def test_multi_error_paths():
    from cfd_amd._capi import CfdError
    from cfd_amd.multi import MultiGraph
    params, fields, active = small_case(6, 5, 1, 65)
    with pytest.raises(CfdError):
        MultiGraph(params, fields, active, 7, devices=[0] * 7)             # more slabs than planes
    with pytest.raises(CfdError):
        MultiGraph(params, fields, active, 2, devices=[0, 0], transport="rccl")    # RCCL: one slab per device
    with pytest.raises(CfdError):
        MultiGraph(params, fields, active, 2, devices=[0, 99])


def test_compiled_host_over_three_slabs_prints_the_reference_stdout():
    """driver/cfd_main --slabs 3: the reference's loop over cfd_multi_* (C++, no Python in the loop), the shipped
    case cut into three slabs on this GPU: the same five lines per step as the golden transcript."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "driver", "cfd_main")
    assert os.path.exists(exe), "build it with make -C driver (done by __graft_entry__.build())"
    out = subprocess.run([exe, "5", "--slabs", "3"], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stderr
    got = out.stdout.splitlines()
    want = open(os.path.join(root, "tests", "golden", "shipped_case_stdout_5steps.txt")).read().splitlines()
    assert got[0] == "starting ..... " and len(got) - 1 == len(want)
    for a, b in zip(got[1:], want):
        if a.startswith("totalDensity: "):
            x, y = float(a.split()[1]), float(b.split()[1])
            assert abs(x - y) <= 1e-12 * abs(y), (a, b)
        else:
            assert a == b


class _VirtualRank:
    """Rank `rank` of a K-slab decomposition run entirely on this GPU by the library's slab driver (device
    copies in place of RCCL), seen the way the pre-flight sees one rank's slab."""

    def __init__(self, name, world, rank, flags, spoil=False):
        from cfd_amd import preflight
        params, fields, active = next((p, f, m) for n, p, f, m in preflight.grids(world, 1) if n == name)
        self.m = _multi(dict(params, dt=1e-7), fields, active, flags, world)
        nx, plane = params["nx"], params["ny"]
        self.a, self.b = (nx * rank // world) * plane, (nx * (rank + 1) // world) * plane
        self.spoil = spoil

    def step(self, n):
        self.m.step(n)

    def read_fields(self, rel):
        out = [f[self.a:self.b].copy() for f in self.m.read_fields(rel)]
        if self.spoil and rel == 0:
            out[4][7] = np.nextafter(out[4][7], 1e9)        # one Temperature value off by one ulp
        return out

    def total_density(self):
        return self.m.total_density()

    def close(self):
        self.m.close()


@pytest.mark.parametrize("flags", [FAST, 0])
def test_preflight_of_a_decomposed_run(flags):
    """The pre-flight bench.py runs before it times N > 1 ranks, rehearsed with three virtual ranks on this GPU: it
    passes on the library's slab driver and names rank, grid, step, slot, property and plane when a single value of
    a single rank is off by one ulp."""
    from cfd_amd import preflight
    for rank in range(3):
        rep = preflight.run(rank, 3, flags, 1, lambda name, p, f, m, x0, x1: _VirtualRank(name, 3, rank, flags))
        assert rep["ok"] and rep["detail"] is None and len(rep["cases"]) == 2 and rep["steps"] == 3, rep
    rep = preflight.run(1, 3, flags, 1, lambda name, p, f, m, x0, x1: _VirtualRank(name, 3, 1, flags, spoil=True))
    assert not rep["ok"]
    assert "rank 1 of 3" in rep["detail"] and "Temperature differs in 1 cells" in rep["detail"] and "step 1" in rep["detail"]


@pytest.mark.parametrize("n,shape,margin", [(3, (900, 200), None), (2, (700, 530), "40"), (4, (1300, 64), "1")])
def test_slabs_whose_chunks_next_to_a_neighbour_are_shorter(n, shape, margin, monkeypatch):
    """Slabs long enough for several chunks of the one-pass step: the chunks next to a neighbouring slab start late
    (after the exchange) and are planned shorter than the others; the chunk lengths change nothing in the result."""
    if margin is not None:
        monkeypatch.setenv("CFD_TRIO_EDGE_MARGIN", margin)
    params, fields, active = small_case(shape[0], shape[1], 1, 67)
    g = _single(params, fields, active, FAST)
    m = _multi(params, fields, active, FAST, n)
    assert m.info()[1] == 2
    for step in range(3):
        g.step(1)
        m.step(1)
        _assert_same(m, g, "step %d" % step)


@pytest.mark.parametrize("flags,shape,n", [(FAST, (40, 300, 1), 3), (0, (12, 9, 7), 2), (FAST, (21, 18, 19), 4)])
def test_checkpoint_restore_of_a_decomposed_run_is_bitwise(flags, shape, n):
    """Four steps on n slabs, checkpoint (three slots + time counters), restore into a NEW slab handle created from
    unrelated fields, three more steps: bit-identical, slot for slot, to seven straight undecomposed steps."""
    params, fields, active = small_case(shape[0], shape[1], shape[2], 71)
    ref = _single(params, fields, active, flags)
    ref.step(7)
    a = _multi(params, fields, active, flags, n)
    a.step(4)
    state = a.checkpoint()
    a.close()
    junk = [np.full_like(np.asarray(f, dtype=np.float64), 1.0 + k) for k, f in enumerate(fields)]
    if shape[2] == 1:
        junk[2][:] = 0.0        # the 2-D form holds no W
    c = _multi(params, junk, active, flags, n)
    c.restore(state)
    assert (c.currModTime, c.currAbsTime) == (4 % 3, 4)
    c.step(3)
    _assert_same(c, ref, "restored")


@pytest.mark.parametrize("flags,shape,n,per_step", [(FAST, (40, 300, 1), 3, 1), (0, (12, 9, 7), 2, 3), (FAST, (4, 33, 1), 2, 2)])
def test_exchange_timing_counts_the_exchanges_of_a_step(flags, shape, n, per_step):
    """The measurement hook of the slab driver: one exchange per step where the step is one pass, two where its tail
    is, three otherwise; a positive duration; nothing recorded while it is off; results unchanged by it."""
    params, fields, active = small_case(shape[0], shape[1], shape[2], 73)
    g = _single(params, fields, active, flags)
    m = _multi(params, fields, active, flags, n)
    m.step(1)
    assert m.exchange_timing() == (0.0, 0)
    m.set_exchange_timing(True)
    m.step(4)
    ms, count = m.exchange_timing(reset=True)
    assert count == 4 * per_step and ms > 0.0, (ms, count)
    m.set_exchange_timing(False)
    m.step(1)
    assert m.exchange_timing() == (0.0, 0)
    g.step(6)
    _assert_same(m, g, "after timed steps")


class _Span:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


@pytest.mark.parametrize("flags,shape", [(FAST, (40, 300, 1)), (FAST, (4, 33, 1)), (FAST, (2, 50, 1)), (0, (9, 40, 1)),
                                         (0, (12, 9, 7)), (FAST, (11, 18, 19))])
def test_rccl_loopback_rank_equals_a_slab_exchanged_by_hand(flags, shape, monkeypatch):
    """The rank form of the slab driver with the slab as its own lower and upper neighbour (CFD_MULTI_LOOPBACK: grouped
    ncclSend / ncclRecv to itself on the side stream, the calls a rank between two others makes, every step form)
    against the same periodic slab stepped through the single-handle API with its ghost planes copied by hand before
    every sweep: all three slots bit-identical after every step."""
    import torch
    from cfd_amd.multi import MultiGraph
    from cfd_amd.solver import AdjGraph
    nx, ny, nz = shape
    params, fields, active = small_case(nx, ny, nz, 79)
    params = dict(params, flags=flags, device=0)
    monkeypatch.setenv("CFD_MULTI_LOOPBACK", "1")
    m = MultiGraph.for_rank(params, fields, active, 0, 1, None)
    monkeypatch.delenv("CFD_MULTI_LOOPBACK")
    assert m.info()[2] is True
    g = AdjGraph(dict(params, ghost_lo=1, ghost_hi=1), fields, active)
    act = np.ascontiguousarray(active, dtype=np.uint8).reshape(nx, ny * nz)

    def fill(depth):
        g.sync()
        (slo, shi, rlo, rhi), nbytes = g.halo_planes(depth)
        t = lambda ptr: torch.as_tensor(_Span(ptr, nbytes), device="cuda")
        t(rlo).copy_(t(shi))
        t(rhi).copy_(t(slo))
        torch.cuda.synchronize()

    for step in range(3):
        m.step(1)
        for sweep in range(3):      # by hand: an exchange before every sweep, the filter with the last one
            fill(1)
            if sweep < 2:
                g.sweep(sweep == 0)
            else:
                g.sweep_and_filter()
        assert (m.currModTime, m.currAbsTime) == (g.currModTime, g.currAbsTime)
        for rel in range(3):
            fm, fg = m.read_fields(rel), g.read_fields(rel)
            for k in range(7):
                assert bits_equal(fm[k], fg[k]), (step, rel, PROPS[k], float(np.max(np.abs(fm[k] - fg[k]))))
    ms, n = m.exchange_timing()
    assert n == 0
