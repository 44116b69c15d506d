"""Pre-flight of a slab-decomposed run: before anything is timed, every rank runs the SAME small global grids
twice -- undecomposed on its own GPU, and as its slab of the decomposition through the driver under test -- and
compares its planes bit for bit after every step, all three time slots, plus the all-reduced total density.

The update reads a 3 x 3 (x 3) neighbourhood per sweep and the cells of a sweep are independent
(applyDiffEq_adj, src/Main.hs:202-226 of the reference, maps over the nodes of one immutable snapshot, :211), so
a decomposition with correctly exchanged ghost planes reproduces the undecomposed fields exactly; only the
density sum (totalDensity, :269-270) depends on the order of summation.  A wrong halo, a wrong neighbour mask
or a stale ghost plane shows up here as a named rank / slot / property / plane instead of as a plausible
throughput number.
"""
import numpy as np

from . import cases
from .solver import AdjGraph

PROPS = ("U", "V", "W", "Density", "Temperature", "Mew", "Dye")


def grids(world, nz):
    """(name, params, fields, active) of the small global grids: x-planes in proportion to the ranks, so that every
    slab owns enough planes for every form of a step (one-pass step: >= 3)."""
    if nz > 1:
        return [("synthetic_3d",) + tuple(cases.synthetic_3d(12 * world, 20, 18, seed=5))]
    return [("synthetic_2d",) + tuple(cases.synthetic_2d(40 * world, 528, seed=11)),
            ("cavity_2d",) + tuple(cases.cavity_2d(24 * world, 300))]


def slab_range(nx, rank, world):
    return nx * rank // world, nx * (rank + 1) // world


def first_difference(got, want, x0, plane):
    """None if the arrays (this rank's planes) are bit-identical, else a description of the first difference."""
    for k in range(7):
        a = np.ascontiguousarray(got[k], dtype=np.float64).view(np.uint64)
        b = np.ascontiguousarray(want[k], dtype=np.float64).view(np.uint64)
        # +0 and -0 are not told apart (the strict kernels' sign of an exact zero is not the reference's)
        bad = np.nonzero((a != b) & ~(((a | b) << np.uint64(1)) == 0))[0]
        if bad.size:
            i = int(bad[0])
            return "%s differs in %d cells, first in plane %d cell %d: %r vs %r" % (
                PROPS[k], bad.size, x0 + i // plane, i % plane, float(got[k][i]), float(want[k][i]))
    return None


def run(rank, world, flags, nz, make_slab, device=-1, steps=3, dt=1e-7, make_ref=None):
    """make_slab(name, params, fields, active, x0, x1): this rank's slab of the decomposed run of grid `name` (params /
    fields / active of planes x0 .. x1-1 only) with step(n), read_fields(slot_rel), total_density() (summed over
    all ranks) and close().  make_ref(params, fields, active): the undecomposed run with the same verbs (default: the
    library on this rank's GPU).  Returns {"ok", "cases", "steps", "detail"}."""
    report = {"ok": True, "steps": steps, "cases": [], "detail": None}
    for name, params, fields, active in grids(world, nz):
        params = dict(params, dt=dt, flags=flags, device=device)
        nx, plane = params["nx"], params["ny"] * params["nz"]
        x0, x1 = slab_range(nx, rank, world)
        ref = (make_ref or AdjGraph)(params, fields, active)
        sl = make_slab(name, dict(params, nx=x1 - x0), [np.ascontiguousarray(f[x0 * plane:x1 * plane]) for f in fields],
                       np.ascontiguousarray(active[x0 * plane:x1 * plane]), x0, x1)
        detail = None
        for step in range(steps):
            ref.step(1)
            sl.step(1)
            for rel in range(3):
                want = [f[x0 * plane:x1 * plane] for f in ref.read_fields(rel)]
                d = first_difference(sl.read_fields(rel), want, x0, plane)
                if d and not detail:
                    detail = "%s, rank %d of %d (planes %d..%d), step %d, slot -%d: %s" % (
                        name, rank, world, x0, x1 - 1, step + 1, rel, d)
            a, b = sl.total_density(), ref.total_density()
            if not abs(a - b) <= 1e-12 * abs(b) and not detail:
                detail = "%s, rank %d, step %d: totalDensity %r vs %r undecomposed" % (name, rank, step + 1, a, b)
        sl.close()
        ref.close()
        report["cases"].append("%s %dx%dx%d" % (name, nx, params["ny"], params["nz"]))
        if detail:
            report["ok"] = False
            report["detail"] = report["detail"] or detail
    return report
