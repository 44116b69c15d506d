"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.

Bars (stated per test):
  * strict arithmetic vs Oracle B in t*sqrt(t) mode: BIT-EXACT (the kernels keep the reference's
    operation order; +0/-0 are not distinguished);
  * vs Oracle B in libm-pow mode (what GHC's ** calls): <= 1e-13 field-scale relative per sweep;
  * totalDensity: <= 1e-12 relative (tree reduction vs sequential fold).
"""
import numpy as np
import pytest

from helpers import PROPS, bits_equal, own_scale_errors, field_errors, oracle_params, rel_err, small_case

pytestmark = pytest.mark.gpu


def _gpu_graph(params, fields, active, flags=0):
    from cfd_amd.solver import AdjGraph
    p = dict(params)
    p["flags"] = flags
    return AdjGraph(p, fields, active)


def _oracle(params, fields, active, pow_mode):
    from oracle import OracleB
    return OracleB(oracle_params(params), fields, active, pow_mode=pow_mode, nthreads=4)


def _assert_slots_bit_equal(g, b, what):
    for rel in range(3):
        fg, fb = g.read_fields(rel), b.read_fields(rel)
        for k in range(7):
            assert bits_equal(fg[k], fb[k]), "%s: slot -%d %s differs, max abs %.3e" % (
                what, rel, PROPS[k], np.max(np.abs(fg[k] - fb[k])))


@pytest.mark.parametrize("shape,seed,zero_v", [((12, 9), 1, False), ((31, 17), 2, True), ((64, 70), 3, False),
                                                ((5, 130), 4, False), ((3, 3), 5, False), ((1, 1), 6, False),
                                                ((2, 200), 7, True), ((70, 300), 8, False),
                                                ((37, 254), 9, False), ((33, 255), 10, False),
                                                ((9, 509), 11, False)])
@pytest.mark.parametrize("flags,chunk,variant", [(1, 32, 8), (0, 32, 12), (0, 1, 13), (0, 5, 14), (0, 32, 8), (0, 1, 8),
                                                 (0, 4, 12), (0, 7, 0), (0, 32, -1), (0, 1, -1)])
def test_sweeps_bit_exact_2d(shape, seed, zero_v, flags, chunk, variant, monkeypatch):
    monkeypatch.setenv("CFD_CHUNK", str(chunk))
    monkeypatch.setenv("CFD_VARIANT", str(variant))
    params, fields, active = small_case(shape[0], shape[1], 1, seed, zero_v)
    g = _gpu_graph(params, fields, active, flags)
    b = _oracle(params, fields, active, pow_mode=1)
    assert g.live_fields == 5
    assert g.kernel_name() == ("sweep2d_simple<strict>" if flags & 1 else "sweep2d_march2<strict>")
    for i, st in enumerate([True, False, False, True, False]):
        g.sweep(st)
        b.sweep(st)
        assert (g.currModTime, g.currAbsTime) == b.time()
        _assert_slots_bit_equal(g, b, "sweep %d" % i)


@pytest.mark.parametrize("flags", [0, 1])
def test_steps_filter_and_density_2d(flags):
    params, fields, active = small_case(40, 33, 1, 11)
    g = _gpu_graph(params, fields, active, flags)
    b = _oracle(params, fields, active, pow_mode=1)
    for step in range(5):
        g.step(1)
        b.step(1)
        assert (g.currModTime, g.currAbsTime) == b.time()
        _assert_slots_bit_equal(g, b, "step %d" % step)
        td, tb = g.total_density(), b.total_density()
        assert abs(td - tb) <= 1e-12 * abs(tb)


def test_single_sweep_vs_libm_pow_form():
    """One sweep from identical state against the oracle in libm-pow mode (GHC's **):
    <= 1e-13 field-scale relative.  (The t*sqrt(t) form differs from pow by a few ulp of mu.)"""
    params, fields, active = small_case(40, 33, 1, 13)
    g = _gpu_graph(params, fields, active, 1)
    b0 = _oracle(params, fields, active, pow_mode=0)
    g.sweep(True)
    b0.sweep(True)
    fg, f0 = g.read_fields(0), b0.read_fields(0)
    for k in range(7):
        assert rel_err(fg[k], f0[k]) <= 1e-13, PROPS[k]


@pytest.mark.parametrize("flags,shape", [(0, (70, 300, 1)), (4, (70, 300, 1)), (4, (37, 29, 35)), (0, (12, 9, 7)),
                                         (0, (37, 29, 35))])
def test_fused_filter_equals_separate_filter(flags, shape):
    """cfd_step (filter fused into the third sweep where the kernel supports it) against sweep, sweep,
    sweep, filter launched separately: bit-identical in all three slots, for both arithmetic forms,
    2-D and 3-D."""
    params, fields, active = small_case(shape[0], shape[1], shape[2], 17)
    ga = _gpu_graph(params, fields, active, flags)
    gb = _gpu_graph(params, fields, active, flags)
    for step in range(4):
        ga.step(1)
        gb.sweep(True); gb.sweep(False); gb.sweep(False); gb.time_filter()
        for rel in range(3):
            fa, fb = ga.read_fields(rel), gb.read_fields(rel)
            for k in range(7):
                assert bits_equal(fa[k], fb[k]), (step, rel, PROPS[k])


@pytest.mark.parametrize("chunk", [1, 2, 5, 64])
@pytest.mark.parametrize("shape", [(1, 1), (2, 5), (3, 300), (4, 1), (70, 241), (129, 37), (33, 480)])
def test_two_sweeps_in_one_pass_equal_two_launches(shape, chunk, monkeypatch):
    """cfd_sweep_pair_and_filter (second and third sweep of a step in one kernel, the once-swept field
    kept on chip) against cfd_sweep(0) + cfd_sweep_and_filter: bit-identical in all three slots, from the
    first step (no filter yet) on; shapes with one row, one column, ragged tiles and several chunks,
    inactive cells included."""
    monkeypatch.setenv("CFD_PAIR_CHUNK", str(chunk))
    monkeypatch.setenv("CFD_PAIR", "1")
    params, fields, active = small_case(shape[0], shape[1], 1, 23)
    ga = _gpu_graph(params, fields, active, 4)
    monkeypatch.setenv("CFD_PAIR", "0")
    gb = _gpu_graph(params, fields, active, 4)
    for step in range(3):
        ga.sweep(True); ga.sweep_pair_and_filter()
        gb.sweep(True); gb.sweep(False); gb.sweep_and_filter()
        assert ga.kernel_timing(2, reset=False)[1] == 0        # timing is off: nothing recorded
        for rel in range(3):
            fa, fb = ga.read_fields(rel), gb.read_fields(rel)
            for k in range(7):
                assert bits_equal(fa[k], fb[k]), (step, rel, PROPS[k])
    # and the launch really was the one-pass kernel
    ga.set_timing(True)
    ga.sweep(True); ga.sweep_pair_and_filter(); ga.sync()
    assert ga.kernel_timing(2)[1] == 1 and ga.kernel_timing(1)[1] == 0


@pytest.mark.parametrize("nts", [256, 128])
@pytest.mark.parametrize("chunk", [1, 2, 3, 7, 64])
@pytest.mark.parametrize("shape", [(1, 1), (2, 5), (3, 300), (4, 1), (5, 17), (70, 241), (129, 37), (33, 480),
                                   (40, 256), (23, 250)])
def test_whole_step_in_one_pass_equals_three_launches(shape, chunk, nts, monkeypatch):
    """cfd_step as ONE launch (all three sweeps of a step and the filter in one pass over memory, one group of
    waves per sweep, the intermediate fields in LDS only) against cfd_sweep(1), cfd_sweep(0), cfd_sweep_and_filter:
    bit-identical in all three slots from the first step (no filter yet) on; one row, one column, ragged and
    several tiles, chunks shorter than the distance between the stages, inactive cells included."""
    monkeypatch.setenv("CFD_TRIO_CHUNK", str(chunk))
    monkeypatch.setenv("CFD_TRIO_NTS", str(nts))
    monkeypatch.setenv("CFD_TRIO", "1")
    params, fields, active = small_case(shape[0], shape[1], 1, 31)
    ga = _gpu_graph(params, fields, active, 4)
    monkeypatch.setenv("CFD_PAIR", "0")
    gb = _gpu_graph(params, fields, active, 4)
    for step in range(4):
        ga.step(1)
        gb.sweep(True); gb.sweep(False); gb.sweep_and_filter()
        assert (ga.currModTime, ga.currAbsTime) == (gb.currModTime, gb.currAbsTime)
        for rel in range(3):
            fa, fb = ga.read_fields(rel), gb.read_fields(rel)
            for k in range(7):
                assert bits_equal(fa[k], fb[k]), (step, rel, PROPS[k], float(np.max(np.abs(fa[k] - fb[k]))))
    # and the launch really was the one-pass kernel
    ga.set_timing(True)
    ga.step(1); ga.sync()
    assert ga.kernel_timing(3)[1] == 1 and ga.kernel_timing(0)[1] == 0 and ga.kernel_timing(2)[1] == 0


def test_whole_step_in_one_pass_default_chunk_wide_grid():
    """The same at the benchmark's width with the chunk length the library picks, plus the cavity walls (U = V = 0
    exactly in the interior: the coefficient rule's other branch in every cell)."""
    import os
    from cfd_amd import cases
    for params, fields, active in (small_case(300, 4096, 1, 37), cases.cavity_2d(96, 700)):
        params = dict(params)
        params["dt"] = 1e-7
        ga = _gpu_graph(params, fields, active, 4)
        os.environ["CFD_PAIR"] = "0"
        try:
            gb = _gpu_graph(params, fields, active, 4)
        finally:
            del os.environ["CFD_PAIR"]
        ga.step(3)
        for _ in range(3):
            gb.sweep(True); gb.sweep(False); gb.sweep_and_filter()
        for rel in range(3):
            fa, fb = ga.read_fields(rel), gb.read_fields(rel)
            for k in range(7):
                assert bits_equal(fa[k], fb[k]), (rel, PROPS[k], float(np.max(np.abs(fa[k] - fb[k]))))


@pytest.mark.parametrize("chunk", [3, 32])
@pytest.mark.parametrize("shape", [(70, 256), (37, 300), (50, 33), (9, 482), (21, 62), (21, 63), (130, 4096)])
def test_narrow_last_tile_shares_blocks_bitwise(shape, chunk, monkeypatch):
    """A row's narrow last tile is handled by single waves, four chunks to a block, in the default streaming
    kernel and in the two-sweeps kernel; a cell's arithmetic does not depend on the tiling, so two steps must
    equal, bit for bit, the same steps run with 64-thread blocks (one wave per block: nothing to share) as
    three launches per step.  Shapes: last tiles of 16, 60, 33, 2, 62 and 63 columns (62 fits the plain sweep's
    wave but not the two-sweeps kernel's, 63 neither) and the benchmark's width; several chunks, the last
    shared block partly empty."""
    params, fields, active = small_case(shape[0], shape[1], 1, 29)
    monkeypatch.setenv("CFD_CHUNK", str(chunk))
    monkeypatch.setenv("CFD_PAIR_CHUNK", str(chunk + 1))
    ga = _gpu_graph(params, fields, active, FAST)
    monkeypatch.setenv("CFD_VARIANT", "22")
    gb = _gpu_graph(params, fields, active, FAST)
    assert ga.pair_available() and not gb.pair_available()
    ga.step(2)
    gb.step(2)
    for rel in range(3):
        fa, fb = ga.read_fields(rel), gb.read_fields(rel)
        for k in range(7):
            assert bits_equal(fa[k], fb[k]), (rel, PROPS[k])


def test_filter_is_noop_until_second_step():
    params, fields, active = small_case(9, 8, 1, 12)
    g = _gpu_graph(params, fields, active, 1)
    g.sweep(True)
    before = g.read_fields(1)
    g.time_filter()
    after = g.read_fields(1)
    for k in range(7):
        assert bits_equal(before[k], after[k])


def test_reference_case_five_steps():
    """The shipped 301x151 case: five steps against Oracle B, and the step-0 totalDensity that
    follows from the reference's constants (45,451 x 0.3 summed left to right)."""
    from cfd_amd import cases
    params, fields, active = cases.reference_case()
    g = _gpu_graph(params, fields, active, 0)
    b = _oracle(params, fields, active, pow_mode=1)
    for step in range(5):
        g.step(1)
        b.step(1)
        if step == 0:
            assert b.total_density() == 13635.299999990304
            assert abs(g.total_density() - 13635.299999990304) <= 1e-12 * 13635.3
        _assert_slots_bit_equal(g, b, "step %d" % step)


def test_tiled_kernel_matches_per_cell_kernel_at_size():
    """Beyond what the CPU oracle reaches in seconds: the tiled kernel against the one-thread-per-cell
    kernel (itself bit-exact against the oracle above) on a 1500 x 1100 grid, three steps, bit-exact."""
    from cfd_amd import cases
    params, fields, active = cases.synthetic_2d(1500, 1100, seed=5)
    ga = _gpu_graph(params, fields, active, 0)
    gb = _gpu_graph(params, fields, active, 1)
    for step in range(3):
        ga.step(1)
        gb.step(1)
        for rel in range(3):
            fa, fb = ga.read_fields(rel), gb.read_fields(rel)
            for k in range(7):
                assert bits_equal(fa[k], fb[k]), (step, rel, PROPS[k])
    assert abs(ga.total_density() - gb.total_density()) == 0.0


@pytest.mark.parametrize("name", ["case_12x9x1", "case_8x7x1_zero_v"])
def test_golden_fixtures_through_c_abi(name):
    """Committed vectors (generated by the literal restatement, libm pow): every sweep from the
    fixture's own previous state agrees to <= 1e-13 field-scale relative; W stays exactly zero."""
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    params = json.loads(str(z["params"]))
    params.update(device=-1, flags=0, ghost_lo=0, ghost_hi=0)
    fields = [z["in_%d" % k] for k in range(7)]
    for i, st in enumerate(z["sweep_flags"]):
        # restart from the fixture's state before sweep i so that differences do not compound
        g = _gpu_graph(params, fields, z["active"], 0)
        g.sweep(bool(st))
        fg = g.read_fields(0)
        for k in range(7):
            assert rel_err(fg[k], z["sweep%d_%d" % (i, k)]) <= 1e-13, (name, i, PROPS[k])
        assert np.all(fg[2] == 0)
        fields = [z["sweep%d_%d" % (i, k)] for k in range(7)]


@pytest.mark.parametrize("world,shape", [(2, (40, 33)), (3, (50, 300)), (4, (9, 20))])
@pytest.mark.parametrize("split,flags", [(False, 0), (True, 0), (True, 4)])
def test_virtual_rank_slabs_match_single_domain(world, shape, split, flags):
    """Slab decomposition on one GPU: `world` handles with ghost planes, halo planes copied between
    them as device tensors that alias the library's memory, against the undecomposed run:
    bit-exact in every owned plane after two steps.  split: each sweep as cfd_sweep_begin (planes
    that need no ghost plane) / exchange / cfd_sweep_end, the order the multi-GPU driver uses, with
    the filter fused into the third sweep."""
    import torch
    from cfd_amd.slab import LocalComm, SlabGraph, split_slabs
    nx, ny = shape
    params, fields, active = small_case(nx, ny, 1, seed=31)
    ref = _gpu_graph(params, fields, active, flags)
    ref.step(2)
    peers = []
    for r, (x0, x1, f, a) in enumerate(split_slabs(fields, active, nx, ny, world)):
        p = dict(params)
        p["nx"] = x1 - x0
        p["flags"] = flags
        peers.append(SlabGraph(p, f, a, rank=r, world=world, comm=LocalComm(r, peers)))
    bounds = [nx * r // world for r in range(world + 1)]
    for _ in range(2):
        for i, st in enumerate((True, False, False)):
            if split:
                for g in peers:
                    g.backend.sweep_begin(st, i == 2)
                for g in peers:
                    g.exchange()
                torch.cuda.synchronize()
                for g in peers:
                    g.backend.sweep_end()
            else:
                for g in peers:
                    g.exchange()
                torch.cuda.synchronize()
                for g in peers:
                    g.backend.sweep(st)
        if not split:
            for g in peers:
                g.time_filter()
    total = 0.0
    for r, g in enumerate(peers):
        for rel in range(3):
            got, want = g.read_fields(rel), ref.read_fields(rel)
            for k in range(7):
                assert bits_equal(got[k], want[k][bounds[r] * ny:bounds[r + 1] * ny]), (r, rel, PROPS[k])
        total += g.backend.total_density()
    assert abs(total - ref.total_density()) <= 1e-12 * abs(total)


@pytest.mark.parametrize("world,shape", [(2, (12, 33)), (3, (15, 300)), (4, (9, 20)), (2, (4, 7)), (5, (71, 500))])
@pytest.mark.parametrize("pair_chunk", [None, 1, 5])
def test_virtual_rank_slabs_two_sweeps_in_one_pass(world, shape, pair_chunk, monkeypatch):
    """Slabs whose step is one sweep behind a one-plane exchange, then the second and third sweep and the
    filter as ONE pass behind a two-plane exchange (cfd_halo_planes_depth 2, the neighbours' masks from
    cfd_set_ghost_active): bit-exact against the undecomposed run in every owned plane and slot.  The
    shapes put inactive cells (the obstacle, scattered cells) on slab boundary planes, so a ghost plane's
    once-swept values depend on the neighbour's mask."""
    import torch
    from cfd_amd.slab import LocalComm, SlabGraph, split_slabs
    if pair_chunk is not None:
        monkeypatch.setenv("CFD_PAIR_CHUNK", str(pair_chunk))
    nx, ny = shape
    params, fields, active = small_case(nx, ny, 1, seed=33)
    rng = np.random.default_rng(5)
    off = rng.random(active.size) < 0.08              # scattered inactive cells, also on boundary planes
    active = active.copy()
    active[off] = 0
    for k in (0, 1):
        fields[k][off] = 0.0
    ref = _gpu_graph(params, fields, active, 4)
    ref.step(3)
    peers = []
    for r, (x0, x1, f, a) in enumerate(split_slabs(fields, active, nx, ny, world)):
        p = dict(params)
        p["nx"] = x1 - x0
        p["flags"] = 4
        peers.append(SlabGraph(p, f, a, rank=r, world=world, comm=LocalComm(r, peers)))
    assert not peers[0].backend.pair_available()        # no masks yet
    assert peers[0].use_pair()                          # hands every slab its neighbours' masks
    assert all(g.backend.pair_available() for g in peers)
    bounds = [nx * r // world for r in range(world + 1)]
    for _ in range(3):
        for g in peers:
            g.exchange()
        torch.cuda.synchronize()
        for g in peers:
            g.backend.sweep(True)
        for g in peers:
            g.comm.exchange(g.backend, depth=2)
        torch.cuda.synchronize()
        for g in peers:
            g.backend.sweep_pair_and_filter()
    total = 0.0
    for r, g in enumerate(peers):
        for rel in range(3):
            got, want = g.read_fields(rel), ref.read_fields(rel)
            for k in range(7):
                assert bits_equal(got[k], want[k][bounds[r] * ny:bounds[r + 1] * ny]), (r, rel, PROPS[k])
        total += g.backend.total_density()
    assert abs(total - ref.total_density()) <= 1e-12 * abs(total)


@pytest.mark.parametrize("world,shape", [(2, (12, 33)), (3, (15, 300)), (4, (9, 20)), (2, (4, 7)), (3, (7, 50)), (5, (71, 500))])
def test_virtual_rank_slabs_edge_planes_in_a_second_launch(world, shape):
    """The first sweep of a step split two planes deep (cfd_sweep_begin_depth / cfd_sweep_end), then the one-pass
    tail: bit-exact against the undecomposed run.  Slabs of 2, 3 and 4 planes have an empty or one-plane interior.
    (The edge planes on a second stream beside the interior launch -- cfd_sweep_end_on -- are exercised by the
    library's own slab driver, tests/test_gpu_multi.py.)"""
    import torch
    from cfd_amd.slab import LocalComm, SlabGraph, split_slabs
    nx, ny = shape
    params, fields, active = small_case(nx, ny, 1, seed=34)
    ref = _gpu_graph(params, fields, active, 4)
    ref.step(3)
    peers = []
    for r, (x0, x1, f, a) in enumerate(split_slabs(fields, active, nx, ny, world)):
        p = dict(params)
        p["nx"] = x1 - x0
        p["flags"] = 4
        peers.append(SlabGraph(p, f, a, rank=r, world=world, comm=LocalComm(r, peers)))
    assert peers[0].use_pair()
    bounds = [nx * r // world for r in range(world + 1)]
    for _ in range(3):
        for g in peers:
            g.exchange()
        for g in peers:
            g.backend.sweep_begin(True, False, edge=2)
        for g in peers:
            g.backend.sweep_end()
        for g in peers:
            g.comm.exchange(g.backend, depth=2)
        for g in peers:
            g.backend.sweep_pair_and_filter()
    for r, g in enumerate(peers):
        for rel in range(3):
            got, want = g.read_fields(rel), ref.read_fields(rel)
            for k in range(7):
                assert bits_equal(got[k], want[k][bounds[r] * ny:bounds[r + 1] * ny]), (r, rel, PROPS[k])
    assert abs(peers[0].total_density() - ref.total_density()) <= 1e-12 * abs(ref.total_density())


def test_split_sweep_with_filter_stays_on_the_own_stream():
    """cfd_sweep_end_on refuses another stream for a sweep begun with the filter (an unfused filter launch would
    not be ordered after it)."""
    import torch
    from cfd_amd._capi import CfdError
    from cfd_amd.solver import AdjGraph
    params, fields, active = small_case(8, 9, 1, seed=2)
    p = dict(params)
    p.update(flags=4, ghost_lo=1, ghost_hi=1)
    g = AdjGraph(p, fields, active)
    side = torch.cuda.Stream()
    g.sweep_begin(False, True)
    with pytest.raises(CfdError):
        g.sweep_end(side.cuda_stream)
    g.sweep_end()
    g.sync()
    g.close()


def test_slab_pair_needs_masks_and_two_planes():
    """A slab without its neighbours' masks, or with a single plane, cannot run the one-pass tail: the
    entry point says so instead of running two sweeps without the exchange they need in between."""
    from cfd_amd._capi import CfdError
    from cfd_amd.solver import AdjGraph
    params, fields, active = small_case(6, 9, 1, seed=2)
    p = dict(params)
    p.update(flags=4, ghost_lo=1, ghost_hi=0)
    g = AdjGraph(p, fields, active)
    assert not g.pair_available()
    with pytest.raises(CfdError):
        g.sweep_pair_and_filter()
    with pytest.raises(CfdError):
        g.set_ghost_active(None, None)                  # a mask is needed on the side that has a neighbour
    g.set_ghost_active(np.ones(9, np.uint8), None)
    assert g.pair_available()
    # the one-pass step needs the masks two planes deep
    assert not g.one_pass_step_available()
    with pytest.raises(CfdError):
        g.step_one_pass()
    with pytest.raises(CfdError):
        g.step_one_pass_begin()
    g.set_ghost_active(np.ones(18, np.uint8), None, depth=2)
    assert g.one_pass_step_available() and g.pair_available()
    g.halo_planes(3)
    with pytest.raises(CfdError):
        g.halo_planes(4)
    with pytest.raises(CfdError):
        g.set_ghost_active(np.ones(27, np.uint8), None, depth=3)
    g.step_one_pass_begin()
    with pytest.raises(CfdError):
        g.sweep_end()                                   # the open call is a step, not a sweep
    with pytest.raises(CfdError):
        g.sweep(True)
    g.step_one_pass_end()
    assert (g.currModTime, g.currAbsTime) == (1, 1)
    two = [f[:18] for f in fields]
    g2 = AdjGraph(dict(p, nx=2), two, active[:18])
    g2.set_ghost_active(np.ones(18, np.uint8), None, depth=2)
    assert g2.pair_available() and not g2.one_pass_step_available()      # three planes are sent: they must be its own
    one = [f[:9] for f in fields]
    p1 = dict(p)
    p1["nx"] = 1
    g1 = AdjGraph(p1, one, active[:9])
    g1.set_ghost_active(np.ones(9, np.uint8), None)
    assert not g1.pair_available()
    with pytest.raises(CfdError):
        g1.halo_planes(2)
    # cfd_step on a slab is refused: its ghost planes need an exchange before every sweep
    for slab in (g, g1):
        with pytest.raises(CfdError) as e:
            slab.step(1)
        assert e.value.code == -4 and "slab" in str(e.value)


@pytest.mark.parametrize("kernel,chunk,variant", [(1, 32, -1), (0, 32, -1), (0, 1, 0), (0, 3, 0), (0, 32, 1),
                                                   (0, 4, 3), (0, 5, 2), (0, 32, 9)])
@pytest.mark.parametrize("shape,seed,zero_v,flags", [((7, 6, 5), 1, False, 0), ((3, 3, 3), 2, True, 0),
                                                      ((12, 9, 1), 3, False, 2), ((4, 1, 70), 4, False, 0),
                                                      ((1, 1, 1), 5, False, 2), ((9, 20, 13), 6, True, 0),
                                                      ((20, 3, 1), 7, False, 0), ((5, 31, 30), 8, False, 0),
                                                      ((40, 15, 29), 9, False, 0)])
def test_sweeps_bit_exact_3d(shape, seed, zero_v, flags, kernel, chunk, variant, monkeypatch):
    """General 3-D kernels, per-cell (kernel=1, variant 9) and tiled in several launch shapes and chunk lengths
    (also reached with nz == 1 when W != 0 or when forced): bit-exact against Oracle B over sweeps
    and full steps.  The shapes straddle tile edges (14 and 30 interior columns per tile side)."""
    monkeypatch.setenv("CFD_CHUNK", str(chunk))
    monkeypatch.setenv("CFD_VARIANT", str(variant))
    params, fields, active = small_case(shape[0], shape[1], shape[2], seed, zero_v)
    if shape == (20, 3, 1):
        fields[2][:] = np.random.default_rng(1).uniform(-0.2, 0.2, fields[2].size)   # nz == 1 with W != 0
    g = _gpu_graph(params, fields, active, flags | kernel)
    b = _oracle(params, fields, active, pow_mode=1)
    assert g.live_fields == 6
    # strict default: the tiled kernel; the per-cell one on request
    assert g.kernel_name() == ("sweep3d_simple<strict>" if kernel or variant == 9 else "sweep3d_march<strict>")
    for i, st in enumerate([True, False, False]):
        g.sweep(st)
        b.sweep(st)
        _assert_slots_bit_equal(g, b, "sweep %d" % i)
    g.time_filter(); b.time_filter()
    for step in range(2):
        g.step(1)
        b.step(1)
        _assert_slots_bit_equal(g, b, "step %d" % step)
        td, tb = g.total_density(), b.total_density()
        assert abs(td - tb) <= 1e-12 * abs(tb)


@pytest.mark.parametrize("name", ["case_7x6x5", "case_3x3x3_zero_vw"])
def test_golden_fixtures_3d_through_c_abi(name):
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    params = json.loads(str(z["params"]))
    params.update(device=-1, flags=0, ghost_lo=0, ghost_hi=0)
    fields = [z["in_%d" % k] for k in range(7)]
    for i, st in enumerate(z["sweep_flags"]):
        g = _gpu_graph(params, fields, z["active"], 0)
        g.sweep(bool(st))
        fg = g.read_fields(0)
        for k in range(7):
            assert rel_err(fg[k], z["sweep%d_%d" % (i, k)]) <= 1e-13, (name, i, PROPS[k])
        fields = [z["sweep%d_%d" % (i, k)] for k in range(7)]


@pytest.mark.parametrize("split,flags,world", [(False, 0, 3), (True, 0, 3), (True, 4, 3), (True, 4, 10)])
def test_virtual_rank_slabs_3d(split, flags, world):
    """3-D slabs on one GPU (world = 10: one plane per slab, no interior at all), whole sweeps or
    split ones, strict or fast arithmetic: bit-exact against the undecomposed run."""
    import torch
    from cfd_amd.slab import LocalComm, SlabGraph, split_slabs
    nx, ny, nz = 10, 7, 6
    params, fields, active = small_case(nx, ny, nz, seed=41)
    ref = _gpu_graph(params, fields, active, flags)
    ref.step(2)
    peers = []
    for r, (x0, x1, f, a) in enumerate(split_slabs(fields, active, nx, ny * nz, world)):
        p = dict(params)
        p["nx"] = x1 - x0
        p["flags"] = flags
        peers.append(SlabGraph(p, f, a, rank=r, world=world, comm=LocalComm(r, peers)))
    bounds = [nx * r // world for r in range(world + 1)]
    for _ in range(2):
        for i, st in enumerate((True, False, False)):
            if split:
                for g in peers:
                    g.backend.sweep_begin(st, i == 2)
                for g in peers:
                    g.exchange()
                torch.cuda.synchronize()
                for g in peers:
                    g.backend.sweep_end()
            else:
                for g in peers:
                    g.exchange()
                torch.cuda.synchronize()
                for g in peers:
                    g.backend.sweep(st)
        if not split:
            for g in peers:
                g.time_filter()
    pl = ny * nz
    for r, g in enumerate(peers):
        for rel in range(3):
            got, want = g.read_fields(rel), ref.read_fields(rel)
            for k in range(7):
                assert bits_equal(got[k], want[k][bounds[r] * pl:bounds[r + 1] * pl]), (r, rel, PROPS[k])


FAST = 4


@pytest.mark.parametrize("kernel_flag,variant", [(0, -1), (0, 8), (1, -1)])
@pytest.mark.parametrize("case", ["small", "small_zero_v", "shipped", "synthetic", "cavity"])
def test_fast_arithmetic_within_1e13_per_sweep(case, kernel_flag, variant, monkeypatch):
    """Re-associated arithmetic (CFD_FLAG_FAST_MATH): every sweep, started from the oracle's own
    state, agrees with Oracle B to <= 1e-13 of the field scale (velocity components relative to the
    velocity scale).  Measured on these cases: <= 6e-15."""
    from cfd_amd import cases
    monkeypatch.setenv("CFD_VARIANT", str(variant))     # -1: default (streamed messages), 8: shared products
    params, fields, active = {
        "small": lambda: small_case(40, 33, 1, 13),
        "small_zero_v": lambda: small_case(31, 17, 1, 2, True),
        "shipped": cases.reference_case,
        "synthetic": lambda: cases.synthetic_2d(300, 280, 1),
        "cavity": lambda: cases.cavity_2d(64, 64),
    }[case]()
    b = _oracle(params, fields, active, pow_mode=1)
    worst = 0.0
    for sweep in range(4):
        cur = b.read_fields(0)
        g = _gpu_graph(params, cur, active, FAST | kernel_flag)
        g.sweep(True)
        b.sweep(sweep == 0)
        errs = field_errors(g.read_fields(0), b.read_fields(0))
        rels = own_scale_errors(g.read_fields(0), b.read_fields(0))
        worst = max(worst, max(errs))
        for k in range(7):
            assert errs[k] <= 1e-13, (case, sweep, PROPS[k], errs[k])
            # relative to the property's OWN scale (V against V, not against U): the per-step tolerance
            assert rels[k] <= 1e-12, (case, sweep, PROPS[k], rels[k])
    print("worst per-sweep error", case, worst)


def test_fast_arithmetic_drift_over_steps():
    """Fast against strict arithmetic over 30 steps (90 sweeps) on the GPU, on a configuration the
    explicit scheme is stable for (smooth fields, dt = 1e-7): the difference stays below 1e-12 of
    the field scale (observed ~1e-13; the oracle's own pow-vs-sqrt difference reaches 1e-14 there).
    With the shipped dt = 1e-5 the reference's scheme is itself unstable -- any last-bit difference,
    including libm-pow against t*sqrt(t) inside the oracle, doubles every sweep and the fields
    overflow within ~20 steps -- so drift on that configuration measures the scheme, not the code."""
    from cfd_amd import cases
    params, fields, active = cases.smooth_2d(160, 130, dt=1e-7)
    gs = _gpu_graph(params, fields, active, 0)
    gf = _gpu_graph(params, fields, active, FAST)
    worst = 0.0
    for n in (10, 20):
        gs.step(n)
        gf.step(n)
        for rel in range(3):
            errs = field_errors(gf.read_fields(rel), gs.read_fields(rel))
            worst = max(worst, max(errs))
            assert max(errs) <= 1e-12, (n, rel, errs)
        assert abs(gf.total_density() - gs.total_density()) <= 1e-12 * abs(gs.total_density())
    print("fast-vs-strict drift after 90 sweeps:", worst)


@pytest.mark.parametrize("kernel_flag,variant,chunk", [(0, -1, 32), (1, -1, 32), (0, 4, 3), (0, 5, 32), (0, 10, 1),
                                                        (0, 12, 5), (0, 15, 32), (0, 14, 2)])
@pytest.mark.parametrize("case", ["small", "small_zero_v", "thin", "synthetic"])
def test_fast_arithmetic_3d_within_1e13_per_sweep(case, kernel_flag, variant, chunk, monkeypatch):
    """The 3-D forms of the re-associated arithmetic -- per-cell kernel, tiled kernel with shared
    products, tiled kernel with streamed messages, in several launch shapes and chunk lengths: every
    sweep, started from the oracle's own state, within 1e-13 of the field scale of Oracle B."""
    from cfd_amd import cases
    monkeypatch.setenv("CFD_VARIANT", str(variant))
    monkeypatch.setenv("CFD_CHUNK", str(chunk))
    params, fields, active = {
        "small": lambda: small_case(9, 20, 13, 6),
        "small_zero_v": lambda: small_case(7, 6, 5, 2, True),
        "thin": lambda: small_case(4, 1, 70, 4),
        "synthetic": lambda: cases.synthetic_3d(40, 36, 33),
    }[case]()
    b = _oracle(params, fields, active, pow_mode=1)
    worst = 0.0
    for sweep in range(3):
        cur = b.read_fields(0)
        g = _gpu_graph(params, cur, active, FAST | kernel_flag)
        assert g.kernel_name().startswith("sweep3d_simple<fast>" if kernel_flag else "sweep3d_")
        g.sweep(True)
        b.sweep(sweep == 0)
        errs = field_errors(g.read_fields(0), b.read_fields(0))
        rels = own_scale_errors(g.read_fields(0), b.read_fields(0))
        worst = max(worst, max(errs))
        for k in range(7):
            assert errs[k] <= 1e-13, (case, sweep, PROPS[k], errs[k])
            # relative to the property's OWN scale (V against V, not against U): the per-step tolerance
            assert rels[k] <= 1e-12, (case, sweep, PROPS[k], rels[k])
    print("worst per-sweep error (3-D)", case, worst)


@pytest.mark.parametrize("flags", [0, FAST])
def test_tiled_3d_matches_per_cell_3d_bitwise(flags, monkeypatch):
    """Both 3-D kernels evaluate the same expressions: 70 x 45 x 61 (several tiles and chunks, ragged
    edges), two steps, every slot bit-identical."""
    from cfd_amd import cases
    monkeypatch.setenv("CFD_VARIANT", "5")      # the product-sharing tiled kernel, for both arithmetics
    params, fields, active = cases.synthetic_3d(70, 45, 61)
    ga = _gpu_graph(params, fields, active, flags)
    gb = _gpu_graph(params, fields, active, flags | 1)
    ga.step(2)
    gb.step(2)
    for rel in range(3):
        fa, fb = ga.read_fields(rel), gb.read_fields(rel)
        for k in range(7):
            assert bits_equal(fa[k], fb[k]), (rel, PROPS[k])


@pytest.mark.parametrize("variant,chunk", [(-1, 32), (20, 32), (21, 5), (22, 1), (23, 32), (24, 7), (28, 3), (31, 32)])
@pytest.mark.parametrize("case", ["small", "small_zero_v", "shipped", "cavity"])
def test_streaming_kernel_within_1e13_per_sweep(case, variant, chunk, monkeypatch):
    """The streaming form of the tiled 2-D kernel (messages instead of shared products) in its launch
    shapes: every sweep, started from the oracle's own state, within 1e-13 of the field scale of
    Oracle B; and a fused last sweep equals sweep + separate filter bit for bit."""
    from cfd_amd import cases
    monkeypatch.setenv("CFD_VARIANT", str(variant))
    monkeypatch.setenv("CFD_CHUNK", str(chunk))
    params, fields, active = {
        "small": lambda: small_case(40, 33, 1, 13),
        "small_zero_v": lambda: small_case(31, 17, 1, 2, True),
        "shipped": cases.reference_case,
        "cavity": lambda: cases.cavity_2d(64, 64),
    }[case]()
    b = _oracle(params, fields, active, pow_mode=1)
    for sweep in range(3):
        cur = b.read_fields(0)
        g = _gpu_graph(params, cur, active, FAST)
        assert g.kernel_name() == "sweep2d_stream<fast>"
        g.sweep(True)
        b.sweep(sweep == 0)
        errs = field_errors(g.read_fields(0), b.read_fields(0))
        rels = own_scale_errors(g.read_fields(0), b.read_fields(0))
        for k in range(7):
            assert errs[k] <= 1e-13, (case, sweep, PROPS[k], errs[k])
            assert rels[k] <= 1e-12, (case, sweep, PROPS[k], rels[k])
    ga = _gpu_graph(params, fields, active, FAST)
    gb = _gpu_graph(params, fields, active, FAST)
    for _ in range(2):
        ga.sweep(True); ga.sweep(False); ga.sweep_and_filter()
        gb.sweep(True); gb.sweep(False); gb.sweep(False); gb.time_filter()
    for rel in range(3):
        fa, fb = ga.read_fields(rel), gb.read_fields(rel)
        for k in range(7):
            assert bits_equal(fa[k], fb[k]), (rel, PROPS[k])


def test_streaming_3d_kernel_shapes_agree_bitwise(monkeypatch):
    """The streaming 3-D kernel gives the same bits for every tile shape, ring depth and chunk length
    (70 x 45 x 61, two steps, inactive block inside), and stays within 1e-12 of the strict kernel."""
    from cfd_amd import cases
    params, fields, active = cases.synthetic_3d(70, 45, 61)
    params = dict(params, dt=1e-7)
    outs = []
    for variant, chunk in [(10, 32), (12, 7), (14, 1), (15, 70), (13, 4)]:
        monkeypatch.setenv("CFD_VARIANT", str(variant))
        monkeypatch.setenv("CFD_CHUNK", str(chunk))
        g = _gpu_graph(params, fields, active, FAST)
        assert g.kernel_name() == "sweep3d_stream<fast>"
        g.step(2)
        outs.append([g.read_fields(rel) for rel in range(3)])
    for o in outs[1:]:
        for rel in range(3):
            for k in range(7):
                assert bits_equal(o[rel][k], outs[0][rel][k]), (rel, PROPS[k])
    monkeypatch.setenv("CFD_VARIANT", "-1")
    gs = _gpu_graph(params, fields, active, 0)
    gs.step(2)
    for rel in range(3):
        assert max(field_errors(outs[0][rel], gs.read_fields(rel))) <= 1e-12


def test_fast_tiled_matches_fast_per_cell_bitwise(monkeypatch):
    """The product-sharing tiled kernel and the per-cell kernel evaluate the same fast expressions;
    1200 x 900, two steps, bit-identical."""
    from cfd_amd import cases
    monkeypatch.setenv("CFD_VARIANT", "8")
    params, fields, active = cases.synthetic_2d(1200, 900, seed=6)
    ga = _gpu_graph(params, fields, active, FAST)
    gb = _gpu_graph(params, fields, active, FAST | 1)
    ga.step(2)
    gb.step(2)
    for rel in range(3):
        fa, fb = ga.read_fields(rel), gb.read_fields(rel)
        for k in range(7):
            assert bits_equal(fa[k], fb[k]), (rel, PROPS[k])


@pytest.mark.parametrize("nz,flags", [(1, 0), (1, 1), (5, 0)])
@pytest.mark.parametrize("scale", [1e-300, 1e-305, 1e-312])
def test_strict_bit_exact_with_subnormal_dye(scale, nz, flags):
    """Dye near and inside the subnormal range (what a point source decays to far away): the strict
    kernels still match Oracle B bit for bit over two steps."""
    params, fields, active = small_case(33, 29, nz, 3)
    fields = [f.copy() for f in fields]
    fields[6] = np.abs(fields[6]) * scale + scale
    if nz == 1:
        fields[2][:] = 0.0
    g = _gpu_graph(params, fields, active, flags)
    b = _oracle(params, fields, active, pow_mode=1)
    g.step(2)
    b.step(2)
    _assert_slots_bit_equal(g, b, "subnormal dye")


def test_rccl_transport_on_aliased_halo_planes():
    """The transport the multi-GPU driver uses, as far as one GPU allows: a process group with the nccl
    (RCCL) backend, world size 1; the halo planes of a slab handle -- torch tensors that alias the
    library's device memory -- are exchanged with the rank itself through batch_isend_irecv, once
    synchronously and once with the interior sweep launched between posting and waiting, then an
    all-reduce.  Run in a child process (own process group).  tests/rccl_self.py prints the checks."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "rccl_self.py")], capture_output=True, text=True,
                         timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "after first exchange: True" in out.stdout, out.stdout
    assert "second exchange ok: True" in out.stdout and "finite after sweep: True" in out.stdout, out.stdout
    # the two-plane halo of the one-pass tail and a mask plane (uint8) through the same transport
    assert "two-plane exchange ok: True" in out.stdout and "mask ok: True" in out.stdout, out.stdout
    assert "pair available: True" in out.stdout and "finite after the one-pass tail: True" in out.stdout, out.stdout
    # SlabGraph.step's two N > 1 forms over the same transport (the rank as its own neighbour) agree bit for bit
    assert "slab step forms agree: True forms: [True, False] finite: True" in out.stdout, out.stdout


def test_compiled_host_prints_reference_stdout():
    """driver/cfd_main (C++ over the C ABI, no Python in the loop) runs the shipped case for five
    steps and prints the same blocks as the golden transcript (generated by the literal oracle)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "driver", "cfd_main")
    assert os.path.exists(exe), "build it with make -C driver (done by __graft_entry__.build())"
    out = subprocess.run([exe, "5"], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stderr
    got = out.stdout.splitlines()
    assert got[0] == "starting ..... "
    want = open(os.path.join(root, "tests", "golden", "shipped_case_stdout_5steps.txt")).read().splitlines()
    assert len(got) - 1 == len(want)
    for a, b in zip(got[1:], want):
        if a.startswith("totalDensity: "):
            # the density sum is a tree on the device and a left fold in the reference
            x, y = float(a.split()[1]), float(b.split()[1])
            assert abs(x - y) <= 1e-12 * abs(y), (a, b)
        else:
            assert a == b


def test_compiled_host_prints_string_domain():
    """cfd_main --string-domain 1: after the step block, the Vorticity slice in the reference's stringDomain
    text form (src/Main.hs:349-352) -- equal, character for character, to the Python host's rendering of the
    same case (Haskell `show` formatting implemented twice, C++ and Python)."""
    import os
    import subprocess
    from cfd_amd import cases, output
    from cfd_amd.solver import AdjGraph
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([os.path.join(root, "driver", "cfd_main"), "1", "--string-domain", "1"],
                         capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stderr
    params, fields, active = cases.reference_case()
    g = AdjGraph(params, fields, active)
    g.step(1)
    want = output.string_domain(g.property_slice("Vorticity"))
    head = "\n".join(out.stdout.split("\n")[:7])        # "starting", five lines, the blank-ish separator
    assert out.stdout == head + "\n" + want + "\n"
    assert want.count("\n") == params["nx"]


def test_error_paths():
    from cfd_amd import _capi
    from cfd_amd.solver import AdjGraph
    params, fields, active = small_case(4, 4, 1, 1)
    bad = dict(params)
    bad["nx"] = 0
    with pytest.raises(_capi.CfdError):
        AdjGraph(bad, [f[:0] for f in fields], active[:0])
    g = AdjGraph(params, fields, active)
    with pytest.raises(_capi.CfdError):
        g.read_fields(3)
    with pytest.raises(_capi.CfdError):
        g.sweep_end()                       # nothing open
    with pytest.raises(_capi.CfdError):
        g.sweep_begin(True, True)           # the filter follows a sweep that does not advance time
    for call in (lambda: g.property_slice(10), lambda: g.property_slice(-1), lambda: g.property_slice(0, 1),
                 lambda: g.property_slice(0, -1), lambda: g.property_slice(0, 0, 3)):
        with pytest.raises(_capi.CfdError):
            call()
    g.sweep_begin(True)
    for call in (lambda: g.sweep_begin(False), lambda: g.sweep(True), g.time_filter, g.sweep_and_filter,
                 lambda: g.step(1), lambda: g.property_slice(0)):
        with pytest.raises(_capi.CfdError):
            call()
    g.sweep_end()
    g.sweep(False)


# ---- output slices (adjGraphToGrid, src/Main.hs:339-346) --------------------------------------
@pytest.mark.parametrize("shape,seed,zero_v,flags", [((12, 9, 1), 1, False, 0), ((31, 17, 1), 2, True, 4),
                                                      ((1, 1, 1), 3, False, 0), ((7, 6, 5), 4, False, 0),
                                                      ((9, 20, 13), 5, True, 4), ((20, 3, 1), 6, False, 2),
                                                      ((3, 300, 2), 7, False, 0), ((300, 2, 3), 8, False, 0)])
def test_output_slices_bit_exact(shape, seed, zero_v, flags):
    """All ten Properties (Speed, Vorticity, Pressure, the stored ones, Mew) at several z-levels and of all
    three time slots, after two steps: BIT-EXACT against Oracle B evaluated on the device's own fields (so
    the bar holds for the re-associated sweep arithmetic too: the slice kernel itself keeps the reference's
    operation order)."""
    nx, ny, nz = shape
    params, fields, active = small_case(nx, ny, nz, seed, zero_v)
    g = _gpu_graph(params, fields, active, flags)
    g.step(2)
    for rel in range(3):
        # an oracle whose three slots hold the device's slot `rel`: slot_rel = 0 there is that state
        b = _oracle(params, g.read_fields(rel), active, pow_mode=1)
        for zl in sorted({0, (nz - 1) // 2, nz - 1}):
            for prp in range(10):
                got = g.property_slice(prp, zl, rel)
                assert got.shape == (nx, ny)
                assert bits_equal(got, b.property_slice(prp, zl)), (shape, rel, zl, prp)
    from cfd_amd.solver import OUTPUT_PROPERTIES, adjGraphToGrid
    assert adjGraphToGrid(g, "Vorticity") == g.property_slice(OUTPUT_PROPERTIES.index("Vorticity"),
                                                              (nz - 1) // 2).tolist()


@pytest.mark.parametrize("name", ["case_12x9x1", "case_7x6x5", "case_3x3x3_zero_vw", "case_8x7x1_zero_v"])
def test_output_slices_golden_fixtures_through_c_abi(name):
    """Committed slices from the literal restatement (libm pow): nine Properties bit-exact, Mew within
    1e-15 relative (t*sqrt(t) against pow(t, 1.5))."""
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z, s = np.load(os.path.join(gold, name + ".npz")), np.load(os.path.join(gold, name + "_slices.npz"))
    params = json.loads(str(z["params"]))
    params.update(device=-1, flags=0, ghost_lo=0, ghost_hi=0)
    last = int(s["last_step"])
    g = _gpu_graph(params, [z["step%d_slot0_%d" % (last, k)] for k in range(7)], z["active"], 0)
    for zl in range(params["nz"]):
        for prp in range(10):
            got, want = g.property_slice(prp, zl), s["z%d_p%d" % (zl, prp)]
            if prp == 8:
                assert rel_err(got, want) <= 1e-15, (name, zl)
            else:
                assert bits_equal(got, want), (name, zl, prp)


@pytest.mark.parametrize("world,shape", [(2, (12, 9, 1)), (3, (11, 40, 3))])
def test_output_slices_of_slabs_match_single_domain(world, shape):
    """Vorticity reaches one plane into the neighbouring slab: each virtual rank's rows (after the ghost
    exchange SlabGraph.property_slice does) equal the undecomposed slice bit for bit."""
    from cfd_amd.slab import LocalComm, SlabGraph, split_slabs
    nx, ny, nz = shape
    params, fields, active = small_case(nx, ny, nz, seed=41)
    ref = _gpu_graph(params, fields, active, 0)
    peers = []
    for r, (x0, x1, f, a) in enumerate(split_slabs(fields, active, nx, ny * nz, world)):
        p = dict(params)
        p["nx"] = x1 - x0
        peers.append(SlabGraph(p, f, a, rank=r, world=world, comm=LocalComm(r, peers)))
    bounds = [nx * r // world for r in range(world + 1)]
    for prp in range(10):
        want = ref.property_slice(prp)
        for r, g in enumerate(peers):
            assert bits_equal(g.property_slice(prp), want[bounds[r]:bounds[r + 1]]), (r, prp)


# ---- the benchmark grid itself (4096 x 4096), through size-independent properties --------------
FULL = 4096


def _full_size_case(seed, vary):
    """Bench-recipe values on the full grid, varying along one axis only ('x' or 'y'); no inflow column or
    obstacle, so that the other axis is a symmetry of the problem."""
    from cfd_amd import cases
    from cfd_amd.solver import make_params
    rng = np.random.default_rng(seed)
    u = lambda: rng.uniform(-1, 1, FULL)
    line = [10 + 0.1 * u(), 0.1 * u(), np.zeros(FULL), 0.3 * (1 + 1e-3 * u()), 223 * (1 + 1e-3 * u()),
            np.full(FULL, 0.1), rng.uniform(0, 1, FULL)]
    line[6][::7] = 0.0                                           # exact zeros: the NaN -> 1 coefficient rule
    shape = (FULL, 1) if vary == "x" else (1, FULL)
    fields = [np.ascontiguousarray(np.broadcast_to(f.reshape(shape), (FULL, FULL))).reshape(-1) for f in line]
    params = make_params(FULL, FULL, 1, cases.REF_DX, cases.REF_DY, cases.REF_DZ, dt=1e-7)
    return params, fields, np.ones(FULL * FULL, np.uint8)


@pytest.mark.parametrize("vary", ["x", "y"])
@pytest.mark.parametrize("flags", [4, 0])
def test_full_size_symmetry_is_kept_bitwise(vary, flags):
    """Fields that vary along one axis only keep that symmetry exactly: after two steps on the 4096 x 4096
    grid every line along the other axis is constant, bit for bit, in all three slots.  Each tile, chunk and
    halo lane of the launch (plain sweep, two-sweeps kernel, fused filter) computes the same numbers from the
    same inputs, so any indexing slip at a tile or chunk edge of the full grid shows up as a broken symmetry.
    (Clamp-to-self at the domain ends preserves it: the clamped neighbour equals the cell.)"""
    params, fields, active = _full_size_case(11, vary)
    g = _gpu_graph(params, fields, active, flags)
    g.step(2)
    for rel in range(3):
        for k, f in enumerate(g.read_fields(rel)):
            a = f.reshape(FULL, FULL).view(np.uint64)
            same = (a == a[:, :1]).all() if vary == "x" else (a == a[:1, :]).all()
            assert same, (vary, flags, rel, PROPS[k])
    # and the one-dimensional profile that is left equals a narrow grid's (64 cells along the symmetric axis)
    n = 64
    shape = (FULL, n) if vary == "x" else (n, FULL)
    small = [np.ascontiguousarray(f.reshape(FULL, FULL)[:shape[0], :shape[1]]).reshape(-1) for f in fields]
    p = dict(params, nx=shape[0], ny=shape[1])
    gs = _gpu_graph(p, small, np.ones(shape[0] * shape[1], np.uint8), flags)
    gs.step(2)
    for rel in range(3):
        big, sm = g.read_fields(rel), gs.read_fields(rel)
        for k in range(7):
            b2 = big[k].reshape(FULL, FULL)[:shape[0], :shape[1]]
            assert bits_equal(b2, sm[k].reshape(shape)), (vary, flags, rel, PROPS[k])


def test_full_size_default_path_against_per_cell_kernel():
    """The benchmark workload itself (synthetic 4096 x 4096, seed 1, dt 1e-7), one step: the strict tiled kernel
    equals the strict one-thread-per-cell kernel bit for bit; the default re-associated path (plain sweep +
    two-sweeps kernel) stays within 3e-13 of the field scale of it (three sweeps at <= 1e-13 each)."""
    from cfd_amd import cases
    params, fields, active = cases.synthetic_2d(FULL, FULL, seed=1)
    params["dt"] = 1e-7
    ref = _gpu_graph(params, fields, active, 1)          # per-cell kernel, strict
    ref.step(1)
    want = [ref.read_fields(rel) for rel in range(3)]
    ref.close()
    g = _gpu_graph(params, fields, active, 0)
    g.step(1)
    for rel in range(3):
        got = g.read_fields(rel)
        for k in range(7):
            assert bits_equal(got[k], want[rel][k]), (rel, PROPS[k])
    g.close()
    g = _gpu_graph(params, fields, active, 4)
    g.step(1)
    for rel in range(3):
        assert max(field_errors(g.read_fields(rel), want[rel])) <= 3e-13, rel
    g.close()


@pytest.mark.parametrize("one_pass_tail", [False, True])
def test_full_size_slabs_equal_undecomposed(one_pass_tail):
    """4096 x 4096 cut into two slabs of 2048 planes (virtual ranks, ghost planes exchanged as aliased device
    tensors) against the undecomposed default path, bit for bit after two steps.  Slab form: three split
    sweeps with the fused filter, or a split sweep and then the one-pass tail behind a two-plane exchange
    (the form the multi-GPU driver takes)."""
    import torch
    from cfd_amd import cases
    from cfd_amd.slab import LocalComm, SlabGraph, split_slabs
    params, fields, active = cases.synthetic_2d(FULL, FULL, seed=2)
    params["dt"] = 1e-7
    params["flags"] = 4
    ref = _gpu_graph(params, fields, active, 4)
    ref.step(2)
    peers = []
    for r, (x0, x1, f, a) in enumerate(split_slabs(fields, active, FULL, FULL, 2)):
        p = dict(params)
        p["nx"] = x1 - x0
        peers.append(SlabGraph(p, f, a, rank=r, world=2, comm=LocalComm(r, peers), pair=one_pass_tail))
    assert peers[0].use_pair() == one_pass_tail
    for _ in range(2):
        for i, st in enumerate((True,) if one_pass_tail else (True, False, False)):
            for g in peers:
                g.backend.sweep_begin(st, i == 2)
            for g in peers:
                g.exchange()
            torch.cuda.synchronize()
            for g in peers:
                g.backend.sweep_end()
        if one_pass_tail:
            for g in peers:
                g.comm.exchange(g.backend, depth=2)
            torch.cuda.synchronize()
            for g in peers:
                g.backend.sweep_pair_and_filter()
    half = FULL // 2 * FULL
    for rel in range(3):
        want = ref.read_fields(rel)
        for r, g in enumerate(peers):
            got = g.read_fields(rel)
            for k in range(7):
                assert bits_equal(got[k], want[k][r * half:(r + 1) * half]), (r, rel, PROPS[k])


@pytest.mark.parametrize("shape,world,one_pass", [((15, 300, 1), 3, True), ((9, 33, 1), 3, True), ((5, 40, 1), 2, False),
                                                  ((12, 9, 7), 2, False)])
def test_torch_slab_driver_takes_the_one_pass_step(shape, world, one_pass):
    """cfd_amd.slab.SlabGraph (the torch.distributed driver a multi-GPU bench falls back to), virtual ranks on this
    GPU: where every slab has three planes and the grid is 2-D with re-associated arithmetic, its step is ONE
    three-plane exchange and the one-pass step; otherwise the earlier forms.  Bit-identical to the undecomposed run."""
    import torch
    from cfd_amd.slab import LocalComm, SlabGraph, split_slabs
    nx, ny, nz = shape
    params, fields, active = small_case(nx, ny, nz, 83)
    params = dict(params, flags=FAST)
    ref = _gpu_graph(params, fields, active, FAST)
    peers = []
    for r, (x0, x1, f, a) in enumerate(split_slabs(fields, active, nx, ny * nz, world)):
        peers.append(SlabGraph(dict(params, nx=x1 - x0), f, a, rank=r, world=world, comm=LocalComm(r, peers)))
    assert [g.use_one_pass() for g in peers] == [one_pass] * world
    for step in range(3):
        ref.step(1)
        if one_pass:
            for g in peers:
                g.comm.exchange(g.backend, depth=3)
            torch.cuda.synchronize()
            for g in peers:
                g.backend.step_one_pass()
        else:
            for g in peers:
                g.use_pair()
            # virtual ranks step one after the other, which is only valid sweep by sweep: exchange all, then sweep all
            for i, st in enumerate((True, False, False)):
                for g in peers:
                    g.exchange()
                torch.cuda.synchronize()
                for g in peers:
                    if i == 2:
                        g.backend.sweep_and_filter()
                    else:
                        g.backend.sweep(st)
        torch.cuda.synchronize()
        x = 0
        for g in peers:
            n = g.backend.planes() * ny * nz
            for rel in range(3):
                got, want = g.read_fields(rel), ref.read_fields(rel)
                for k in range(7):
                    assert bits_equal(got[k], want[k][x:x + n]), (step, rel, PROPS[k])
            x += n


@pytest.mark.timeout(400)
@pytest.mark.parametrize("vary", [0, 2])
def test_full_size_symmetry_3d(vary):
    """512 x 512 x 512 (the 3-D benchmark grid), fields varying along one axis only (x or z): after one step of
    the default re-associated kernels every plane across that axis is constant, bit for bit -- all tiles, halo
    rings and chunks of the full launch compute the same numbers -- and the remaining profile equals that of
    a grid 24 cells wide in the two symmetric directions."""
    from cfd_amd import cases
    from cfd_amd.solver import make_params
    n = 512
    rng = np.random.default_rng(13 + vary)
    u = lambda: rng.uniform(-1, 1, n)
    line = [10 + 0.1 * u(), 0.1 * u(), 0.1 * u(), 0.3 * (1 + 1e-3 * u()), 223 * (1 + 1e-3 * u()),
            np.full(n, 0.1), rng.uniform(0, 1, n)]
    line[6][::5] = 0.0
    bshape = [1, 1, 1]
    bshape[vary] = n

    def grid_of(dims):
        return [np.ascontiguousarray(np.broadcast_to(f.reshape(bshape), dims)).reshape(-1) for f in line]
    dims = (n, n, n)
    params = make_params(n, n, n, cases.REF_DX, cases.REF_DY, cases.REF_DX, dt=1e-7)
    g = _gpu_graph(params, grid_of(dims), np.ones(n ** 3, np.uint8), 4)
    g.step(1)
    sdims = [24, 24, 24]
    sdims[vary] = n
    ps = dict(params, nx=sdims[0], ny=sdims[1], nz=sdims[2])
    gs = _gpu_graph(ps, grid_of(tuple(sdims)), np.ones(sdims[0] * sdims[1] * sdims[2], np.uint8), 4)
    gs.step(1)
    first = [slice(0, 1)] * 3
    first[vary] = slice(None)
    for rel in range(2):
        big, sm = g.read_fields(rel), gs.read_fields(rel)
        for k in range(7):
            a = big[k].reshape(dims).view(np.uint64)
            assert (a == a[tuple(first)]).all(), (vary, rel, PROPS[k])
            prof = big[k].reshape(dims)[tuple(first)].reshape(-1)
            sprof = sm[k].reshape(sdims)[tuple(first)].reshape(-1)
            assert bits_equal(prof, sprof), (vary, rel, PROPS[k])
        del big


@pytest.mark.timeout(400)
def test_strict_tiled_3d_matches_per_cell_at_256_cube():
    """256 x 256 x 256 (19 x 19 tiles of 14 x 14 columns, 8 chunks of 32 planes), reference operation order: one
    whole step of the tiled kernel (own products and left-hand sides through LDS, kept values refetched) against
    the per-cell kernel, which shares none of that plumbing -- every slot bit-identical."""
    from cfd_amd import cases
    n = 256
    params, fields, active = cases.synthetic_3d(n, n, n, seed=5)
    params["dt"] = 1e-7
    ga = _gpu_graph(params, fields, active, 0)
    assert ga.kernel_name() == "sweep3d_march<strict>"
    ga.step(1)
    got = [ga.read_fields(rel) for rel in range(2)]
    ga.close()
    gb = _gpu_graph(params, fields, active, 1)
    assert gb.kernel_name() == "sweep3d_simple<strict>"
    gb.step(1)
    for rel in range(2):
        want = gb.read_fields(rel)
        for k in range(7):
            assert bits_equal(got[rel][k], want[k]), (rel, PROPS[k])


@pytest.mark.parametrize("what,t_scale,rho_scale", [("tiny T", 1e-240, 1.0), ("subnormal-range T", 1e-300, 1.0),
                                                    ("huge T", 1e200, 1.0), ("tiny Density", 1.0, 1e-250),
                                                    ("huge Density", 1.0, 1e250), ("small T", 1e-200, 1.0)])
@pytest.mark.parametrize("shape", [(12, 9, 1), (7, 6, 5)])
def test_strict_kernels_far_outside_the_physical_range(what, t_scale, rho_scale, shape):
    """The strict kernels divide and take square roots of Temperature, Density and viscosity with the compiler's
    sequences minus their operand scaling (sweep_math.h: div_normal, sqrt_normal), which equal IEEE division and
    square root for operands in the middle of the exponent range.  Pinned here well beyond that range (fields scaled
    by 1e-300 .. 1e+250, where products under- and overflow): one sweep still equals the oracle bit for bit."""
    params, fields, active = small_case(shape[0], shape[1], shape[2], seed=3)
    fields = [np.array(f, dtype=np.float64) for f in fields]
    fields[4] *= t_scale
    fields[3] *= rho_scale
    g = _gpu_graph(params, fields, active, 0)
    b = _oracle(params, fields, active, pow_mode=1)
    with np.errstate(all="ignore"):
        g.sweep(True)
        b.sweep(True)
    fg, fb = g.read_fields(0), b.read_fields(0)
    for k in range(7):
        a, c = np.asarray(fg[k]), np.asarray(fb[k])
        both_nan = np.isnan(a) & np.isnan(c)
        assert bits_equal(np.where(both_nan, 0.0, a), np.where(both_nan, 0.0, c)), (what, PROPS[k])
