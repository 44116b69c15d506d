"""GPU parity tests of whole steps: the benchmarked path (cfd_step with the re-associated arithmetic: the whole
step in one launch) against the CPU oracle's step, checkpoint / restore, and the
long-horizon lock-step run of the 8192 x 8192 configuration's recipe on a grid the CPU finishes in seconds.

Bars: strict arithmetic vs Oracle B (t*sqrt(t) mode) BIT-EXACT; re-associated arithmetic <= 1e-13 of the field
scale per sweep (3e-13 per step of three sweeps), velocity components measured against the velocity scale, and
<= 1e-12 per step with every property measured against its OWN largest magnitude (helpers.own_scale_errors: V
against V, not against U; floor 1 % of the velocity scale) or twice the oracle's own pow-vs-sqrt difference;
totalDensity <= 1e-12 relative.  The cell-by-cell relative error is printed, not asserted: where a field passes
through zero it measures the scheme's cancellation noise, not the implementation.
"""
import numpy as np
import pytest

from helpers import PROPS, bits_equal, elem_rel_errors, field_errors, oracle_params, own_scale_errors, small_case

pytestmark = pytest.mark.gpu
FAST = 4


def _graph(params, fields, active, flags):
    from cfd_amd.solver import AdjGraph
    return AdjGraph(dict(params, flags=flags), fields, active)


def _oracle(params, fields, active, pow_mode=1, nthreads=4):
    from oracle import OracleB
    return OracleB(oracle_params(params), fields, active, pow_mode=pow_mode, nthreads=nthreads)


def _case(name):
    from cfd_amd import cases
    return {"synthetic": lambda: cases.synthetic_2d(300, 280, 1),
            "cavity": lambda: cases.cavity_2d(64, 64),
            "small": lambda: small_case(40, 33, 1, 13),
            "small3d": lambda: small_case(21, 18, 19, 5)}[name]()


@pytest.mark.parametrize("case", ["synthetic", "cavity", "small", "small3d"])
def test_fast_step_against_oracle_step(case):
    """cfd_step with CFD_FLAG_FAST_MATH -- the path bench.py times: in 2-D all three sweeps of the step and the
    time filter in one launch -- against OracleB.step.  Every step starts from the oracle's
    own three slots and time counters (cfd_set_time + cfd_write_fields), so each comparison is of ONE step from
    identical state: all three slots, <= 3e-13 of the field scale (three sweeps at 1e-13 each) and, per step,
    <= 1e-12 of every property's own scale (3e-12 for the cavity's V, which is 1e-5 of the lid speed) -- or, where
    the scheme itself amplifies a last-bit change beyond that, <= twice what the oracle's own step moves by when its
    viscosity is evaluated with libm pow instead of t*sqrt(t)."""
    params, fields, active = _case(case)
    b = _oracle(params, fields, active)
    g = _graph(params, fields, active, FAST)
    two_d = params["nz"] == 1
    worst_f, worst_o, worst_e, worst_y = 0.0, 0.0, 0.0, 0.0
    for step in range(4):
        g.restore((b.time()[0], b.time()[1], [b.read_fields(r) for r in range(3)]))
        # the yardstick: the oracle's own step from the same current slot with libm pow (what GHC's ** calls) in
        # place of t*sqrt(t) -- a last-bit change of the viscosity and nothing else
        op = _oracle(params, b.read_fields(0), active, pow_mode=0)
        g.set_timing(True)
        g.step(1)
        g.sync()
        plain, fused, pair, trio = (g.kernel_timing(kind)[1] for kind in range(4))
        g.set_timing(False)
        if two_d:
            assert (plain, fused, pair, trio) == (0, 0, 0, 1)       # the whole step really ran as one launch
        else:
            assert (plain, fused, pair, trio) == ((2, 1, 0, 0) if step >= 1 else (3, 0, 0, 0))
        b.step(1)
        op.step(1)
        yard = own_scale_errors(op.read_fields(0), b.read_fields(0))
        op.close()
        worst_y = max(worst_y, max(yard))
        assert (g.currModTime, g.currAbsTime) == b.time()
        for rel in range(3):
            got, want = g.read_fields(rel), b.read_fields(rel)
            fe, oe, ee = field_errors(got, want), own_scale_errors(got, want), elem_rel_errors(got, want)
            worst_f, worst_o, worst_e = max(worst_f, max(fe)), max(worst_o, max(oe)), max(worst_e, max(ee))
            for k in range(7):
                assert fe[k] <= 3e-13, (case, step, rel, PROPS[k], fe[k])
                # The cavity's V (1e-5 of the lid speed, so measured against the floor of 1 % of the velocity scale):
                # 2.4e-12 of that floor = 2.4e-14 of the velocity scale, the rounding noise of the U-scale terms its
                # equation carries; the yardstick does not cover it in the first step (no flow yet, nothing for the
                # viscosity to act on).  Stated here as the one exception to the 1e-12 bar.
                bar = 3e-12 if (case == "cavity" and k == 1) else 1e-12
                assert oe[k] <= max(bar, 2.0 * yard[k]), (case, step, rel, PROPS[k], oe[k], yard[k])
        td, tb = g.total_density(), b.total_density()
        assert abs(td - tb) <= 1e-12 * abs(tb)
    print("fast step vs oracle step", case, "field-scale %.2e own-scale %.2e (oracle pow-vs-sqrt %.2e) cell-relative "
          "(floor 1e-3) %.2e" % (worst_f, worst_o, worst_y, worst_e))


@pytest.mark.parametrize("flags,shape", [(0, (40, 33, 1)), (FAST, (70, 300, 1)), (FAST, (21, 18, 19)), (0, (9, 8, 7))])
def test_checkpoint_restore_is_bitwise(flags, shape):
    """Four steps, checkpoint (three slots + currModTime + currAbsTime), restore into a NEW handle created
    from unrelated fields, three more steps: bit-identical, slot for slot, to seven straight steps."""
    params, fields, active = small_case(shape[0], shape[1], shape[2], 41)
    ref = _graph(params, fields, active, flags)
    ref.step(7)
    a = _graph(params, fields, active, flags)
    a.step(4)
    state = a.checkpoint()
    a.close()
    junk = [np.full_like(np.asarray(f, dtype=np.float64), 1.0 + k) for k, f in enumerate(fields)]
    if shape[2] == 1:
        junk[2][:] = 0.0        # the 2-D form holds no W
    c = _graph(params, junk, active, flags)
    c.restore(state)
    assert (c.currModTime, c.currAbsTime) == (state[0], state[1]) == (4 % 3, 4)
    c.step(3)
    assert (c.currModTime, c.currAbsTime) == (ref.currModTime, ref.currAbsTime)
    for rel in range(3):
        fa, fb = c.read_fields(rel), ref.read_fields(rel)
        for k in range(7):
            assert bits_equal(fa[k], fb[k]), (rel, PROPS[k])
    assert c.total_density() == ref.total_density()


def test_write_fields_rejects_w_in_the_2d_form():
    from cfd_amd._capi import CfdError
    params, fields, active = small_case(12, 9, 1, 3)
    g = _graph(params, fields, active, 0)
    assert g.live_fields == 5
    bad = [np.array(f, dtype=np.float64) for f in fields]
    bad[2][5] = 1e-300
    with pytest.raises(CfdError) as e:
        g.write_fields(bad, 0)
    assert e.value.code == -4
    with pytest.raises(CfdError):
        g.set_time(3, 0)
    with pytest.raises(CfdError):
        g.write_fields(fields, 3)
    g3 = _graph(params, fields, active, 2)      # CFD_FLAG_FORCE_3D
    g3.write_fields(bad, 0)
    assert g3.read_fields(0)[2][5] == 1e-300


@pytest.mark.timeout(300)
def test_long_horizon_lock_step_256sq():
    """The long-horizon configuration (8192 x 8192, total density every step, drift against the oracle) on its
    own recipe at 256 x 256 (cases.synthetic_2d seed 4, dt = 1e-7), 100 steps = 300 sweeps with the strict
    kernels, the re-associated kernels and the CPU oracle in lock-step:
      * strict == Oracle B bit for bit at every checkpoint, all three slots;
      * re-associated vs strict stays within 25x of what a last-bit change of the viscosity alone does to the
        oracle itself (its libm-pow form against its t*sqrt(t) form: the scheme's own amplification), floor
        1e-12.  Measured ratios at steps 25 / 50 / 75 / 100: 5.8, 6.2, 5.6, 13.7 -- the re-associated form
        perturbs every term of every equation by a last bit, the yardstick only the viscous ones;
      * the three total densities agree to 1e-12 at every step (tree sum vs sequential fold included)."""
    from cfd_amd import cases
    params, fields, active = cases.synthetic_2d(256, 256, seed=4)
    params["dt"] = 1e-7
    gs, gf = _graph(params, fields, active, 0), _graph(params, fields, active, FAST)
    ob, op = _oracle(params, fields, active, 1, 8), _oracle(params, fields, active, 0, 8)
    worst_ratio = 0.0
    for step in range(1, 101):
        gs.step(1); gf.step(1); ob.step(1); op.step(1)
        ds, df, do = gs.total_density(), gf.total_density(), ob.total_density()
        assert abs(ds - do) <= 1e-12 * abs(do) and abs(df - do) <= 1e-12 * abs(do), (step, ds, df, do)
        if step % 25 == 0:
            for rel in range(3):
                fs, fo = gs.read_fields(rel), ob.read_fields(rel)
                for k in range(7):
                    assert bits_equal(fs[k], fo[k]), (step, rel, PROPS[k])
            fo = ob.read_fields(0)
            yard = max(field_errors(op.read_fields(0), fo))
            d_fs = max(field_errors(gf.read_fields(0), fo))
            assert np.all(np.isfinite(gf.read_fields(0)[3]))
            assert d_fs <= max(25.0 * yard, 1e-12), (step, d_fs, yard)
            worst_ratio = max(worst_ratio, d_fs / max(yard, 1e-300))
            print("step %d: oracle pow-vs-sqrt %.2e, fast-vs-oracle %.2e" % (step, yard, d_fs))
    print("worst fast / yardstick ratio", worst_ratio)
