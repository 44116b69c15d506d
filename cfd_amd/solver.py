"""Host-side mirror of the reference's step interface, backed by libcfdhip.so.

The reference drives its update through four functions on an immutable `AdjGraph`
(src/Main.hs:192-270).  The same names are kept here so that a caller -- and the parity
tests -- read like the reference; the graph lives in GPU memory and is updated in place, so
each function returns the graph it was given rather than a new one.

    reference (Haskell)                      here
    -------------------------------------    ------------------------------------------
    initialGrid_adj :: AdjGraph              AdjGraph(params, fields, active)
    runSingleStep g                          runSingleStep(g)
    applyDiffEq_adj calcSteps_adj st g       applyDiffEq_adj(st, g)
    correctPrevTime g calcSteps_adj          correctPrevTime(g)
    totalDensity g                           totalDensity(g)
    currModTime g / currAbsTime g            g.currModTime / g.currAbsTime
    adjGraphToGrid g prp rowLength zLevel    adjGraphToGrid(g, prp, zLevel)
    runTimeSteps_Print                       runTimeSteps_Print(g, nsteps)
"""
import ctypes

import numpy as np

from . import _capi
from ._capi import CfdParams, check

PROPERTIES = ("U", "V", "W", "Density", "Temperature", "Mew", "Dye")   # propIndex order
# data Property (src/SolutionDomain.hs:30), the numbering of output slices
OUTPUT_PROPERTIES = ("Speed", "Vorticity", "Pressure", "U", "V", "W", "Density", "Temperature", "Mew", "Dye")

# reference constants (src/SolutionDomain.hs:237-260, src/Main.hs:55-56, :148)
TIME_STEP = 0.00001
SUTHERLAND_CONSTANT = 120.0
SPECIFIC_HEAT_CV = 716.0
GAS_CONSTANT_R = 1005.0 - 716.0
HEAT_CONDUCTIVITY_K = 0.025
DYE_FACTOR = 0.001


def sutherland_lambda():
    """initialMew*(initialTemperature + S)/(initialTemperature**1.5), src/SolutionDomain.hs:249-251."""
    return (0.1 * (223.0 + SUTHERLAND_CONSTANT)) / (223.0 ** 1.5)


def make_params(nx, ny, nz, dx, dy, dz, dt=TIME_STEP, device=-1, flags=0, ghost_lo=0, ghost_hi=0):
    return dict(nx=int(nx), ny=int(ny), nz=int(nz), dx=float(dx), dy=float(dy), dz=float(dz), dt=float(dt),
                lambda_=sutherland_lambda(), suth=SUTHERLAND_CONSTANT, R=GAS_CONSTANT_R, cv=SPECIFIC_HEAT_CV,
                k=HEAT_CONDUCTIVITY_K, dye_f=DYE_FACTOR, device=int(device), flags=int(flags),
                ghost_lo=int(ghost_lo), ghost_hi=int(ghost_hi))


class AdjGraph:
    """Three time slots of the seven stored properties plus the `active` mask, resident on one GPU."""

    def __init__(self, params, fields, active):
        """params: dict from make_params(); fields: 7 arrays (propIndex order) of nx*ny*nz doubles in
        node order (x slowest, z fastest); active: nx*ny*nz flags."""
        self._lib = _capi.load()
        self.params = dict(params)
        self._p = CfdParams(**self.params)
        self.n = self._p.nx * self._p.ny * self._p.nz
        f = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in fields]
        if len(f) != _capi.CFD_NPROP or any(x.size != self.n for x in f):
            raise ValueError("fields must be 7 arrays of nx*ny*nz values")
        act = np.ascontiguousarray(active, dtype=np.uint8).reshape(-1)
        if act.size != self.n:
            raise ValueError("active must have nx*ny*nz entries")
        self._h = ctypes.c_void_p()
        check(self._lib.cfd_create(ctypes.byref(self._p), _capi.ptr_array(f), act.ctypes.data,
                                   ctypes.byref(self._h)))

    # --- state ---------------------------------------------------------------------------
    def _time(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        check(self._lib.cfd_time(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    @property
    def currModTime(self):
        return self._time()[0]

    @property
    def currAbsTime(self):
        return self._time()[1]

    @property
    def live_fields(self):
        n = ctypes.c_int()
        check(self._lib.cfd_live_fields(self._h, ctypes.byref(n)))
        return n.value

    def read_fields(self, slot_rel=0):
        """Seven host arrays of time slot currModTime - slot_rel (0, 1 or 2 slots back)."""
        out = [np.empty(self.n, dtype=np.float64) for _ in range(_capi.CFD_NPROP)]
        check(self._lib.cfd_read_fields(self._h, int(slot_rel), _capi.ptr_array(out)))
        return out

    def write_fields(self, fields, slot_rel=0):
        """Overwrites time slot currModTime - slot_rel with seven host arrays (the inverse of read_fields)."""
        f = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in fields]
        if len(f) != _capi.CFD_NPROP or any(x.size != self.n for x in f):
            raise ValueError("fields must be 7 arrays of nx*ny*nz values")
        check(self._lib.cfd_write_fields(self._h, int(slot_rel), _capi.ptr_array(f)))
        return self

    def set_time(self, currModTime, currAbsTime):
        check(self._lib.cfd_set_time(self._h, int(currModTime), int(currAbsTime)))
        return self

    def checkpoint(self):
        """Everything a resumed run needs: (currModTime, currAbsTime, [slot 0, slot -1, slot -2])."""
        cmt, cat = self._time()
        return cmt, cat, [self.read_fields(r) for r in range(3)]

    def restore(self, state):
        """Puts a checkpoint() back (into this or another graph of the same shape)."""
        cmt, cat, slots = state
        self.set_time(cmt, cat)
        for r in range(3):
            self.write_fields(slots[r], r)
        return self

    def property_slice(self, prp, z_level=None, slot_rel=0):
        """adjGraphToGrid (src/Main.hs:339-346): output property `prp` (name or index in OUTPUT_PROPERTIES)
        of the cells with z == z_level as an (nx, ny) array; z_level defaults to the reference's
        quot (maxPos Z) 2.  slot_rel = 0 is the slot the reference reads."""
        k = OUTPUT_PROPERTIES.index(prp) if isinstance(prp, str) else int(prp)
        if z_level is None:
            z_level = (self._p.nz - 1) // 2
        out = np.empty((self._p.nx, self._p.ny), dtype=np.float64)
        check(self._lib.cfd_property_slice(self._h, k, int(z_level), int(slot_rel), out.ctypes.data))
        return out

    # --- update ----------------------------------------------------------------------------
    def step(self, nsteps=1):
        check(self._lib.cfd_step(self._h, int(nsteps)))
        return self

    def sweep(self, step_time):
        check(self._lib.cfd_sweep(self._h, int(bool(step_time))))
        return self

    def time_filter(self):
        check(self._lib.cfd_time_filter(self._h))
        return self

    def sweep_and_filter(self):
        """A sweep that does not advance time followed by the time filter (the end of a step)."""
        check(self._lib.cfd_sweep_and_filter(self._h))
        return self

    def sweep_pair_and_filter(self):
        """The second and third sweep of a step and the time filter (one pass over memory where the
        library supports it)."""
        check(self._lib.cfd_sweep_pair_and_filter(self._h))
        return self

    def sweep_begin(self, step_time, with_filter=False, edge=1):
        """First half of a split sweep: the planes that are `edge` or more planes away from a ghost plane
        (edge = 1: all that need no ghost plane; see include/cfd_hip.h)."""
        check(self._lib.cfd_sweep_begin_depth(self._h, 1 if step_time else 0, 1 if with_filter else 0, int(edge)))
        return self

    def sweep_end(self, raw_stream=None):
        """Second half of a split sweep: the planes left out by sweep_begin, then the slot rotation.
        raw_stream: launch them on that HIP stream (hipStream_t value) instead of the graph's own."""
        if raw_stream is None:
            check(self._lib.cfd_sweep_end(self._h))
        else:
            check(self._lib.cfd_sweep_end_on(self._h, ctypes.c_void_p(int(raw_stream))))
        return self

    def total_density(self):
        v = ctypes.c_double()
        check(self._lib.cfd_total_density(self._h, ctypes.byref(v)))
        return v.value

    # --- plumbing --------------------------------------------------------------------------
    def sync(self):
        check(self._lib.cfd_sync(self._h))

    def set_timing(self, enable):
        check(self._lib.cfd_set_timing(self._h, int(bool(enable))))

    def sweep_timing(self, reset=True):
        """(total milliseconds, launches) of the sweep kernel since the last reset."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(self._lib.cfd_sweep_timing(self._h, ctypes.byref(ms), ctypes.byref(n), int(bool(reset))))
        return ms.value, n.value

    def kernel_timing(self, kind, reset=True):
        """(total milliseconds, launches) of kind 0 = plain sweeps, 1 = sweeps fused with the time filter,
        2 = two sweeps and the filter in one pass, 3 = a whole step (three sweeps and the filter) in one pass."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(self._lib.cfd_kernel_timing(self._h, int(kind), ctypes.byref(ms), ctypes.byref(n), int(bool(reset))))
        return ms.value, n.value

    def halo_planes(self, depth=1):
        """Device addresses (ints, 0 for none) of send_lo, send_hi, recv_lo, recv_hi for a halo `depth` planes
        deep, and the size of each range in bytes."""
        p = [ctypes.c_void_p() for _ in range(4)]
        nbytes = ctypes.c_size_t()
        check(self._lib.cfd_halo_planes_depth(self._h, int(depth), *[ctypes.byref(x) for x in p],
                                              ctypes.byref(nbytes)))
        return [x.value or 0 for x in p], nbytes.value

    def set_ghost_active(self, lo_planes, hi_planes, depth=1):
        """Masks of the neighbouring slabs' `depth` (1 or 2) planes next to this slab, each side in ascending x
        (None where there is no neighbour)."""
        def arr(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
        lo, hi = arr(lo_planes), arr(hi_planes)
        plane = self._p.ny * self._p.nz
        for a in (lo, hi):
            if a is not None and a.size != depth * plane:
                raise ValueError("a mask has depth*ny*nz entries")
        check(self._lib.cfd_set_ghost_active_depth(self._h, int(depth), None if lo is None else lo.ctypes.data,
                                                   None if hi is None else hi.ctypes.data))
        return self

    def one_pass_step_available(self):
        """True if a whole step (three sweeps and the filter) runs as one pass over memory for this graph as it is now."""
        v = ctypes.c_int()
        check(self._lib.cfd_one_pass_step_available(self._h, ctypes.byref(v)))
        return bool(v.value)

    def step_one_pass(self):
        """runSingleStep as one launch (a slab: after a three-plane exchange)."""
        check(self._lib.cfd_step_one_pass(self._h))
        return self

    def step_one_pass_begin(self):
        check(self._lib.cfd_step_one_pass_begin(self._h))
        return self

    def step_one_pass_end(self, stream=None):
        check(self._lib.cfd_step_one_pass_end_on(self._h, ctypes.c_void_p(self.stream() if stream is None else stream)))
        return self

    def pair_available(self):
        """True if sweep_pair_and_filter runs as one pass over memory for this graph as it is now."""
        v = ctypes.c_int()
        check(self._lib.cfd_pair_available(self._h, ctypes.byref(v)))
        return bool(v.value)

    def stream(self):
        s = ctypes.c_void_p()
        check(self._lib.cfd_stream(self._h, ctypes.byref(s)))
        return s.value or 0

    def set_stream(self, raw_stream):
        """Launch on a caller-owned HIP stream (raw hipStream_t value, 0 = default stream)."""
        check(self._lib.cfd_set_stream(self._h, ctypes.c_void_p(int(raw_stream))))

    def kernel_name(self):
        return self._lib.cfd_kernel_name(self._h).decode()

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.cfd_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --- the reference's verbs (src/Main.hs) -------------------------------------------------------
def runSingleStep(g):
    """src/Main.hs:192-200."""
    return g.step(1)


def applyDiffEq_adj(stepTime, g):
    """src/Main.hs:202-226 with the fixed equation schedule calcSteps_adj (src/Main.hs:180-187)."""
    return g.sweep(stepTime)


def correctPrevTime(g):
    """src/Main.hs:228-258."""
    return g.time_filter()


def totalDensity(g):
    """src/Main.hs:269-270."""
    return g.total_density()


def adjGraphToGrid(g, prp, zLevel=None):
    """src/Main.hs:339-346 (rowLength is always ny): list of nx rows of ny values."""
    return g.property_slice(prp, zLevel).tolist()


def _hs_show_double(v):
    """Haskell `show` of a Double: shortest round-trip digits, positional for 0.1 <= |v| < 10^7,
    otherwise d.ddde<n>."""
    v = float(v)
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "Infinity" if v > 0 else "-Infinity"
    if v == 0:
        return "-0.0" if np.signbit(v) else "0.0"
    if 0.1 <= abs(v) < 1e7:
        return np.format_float_positional(v, unique=True, trim="0")
    m, _, e = np.format_float_scientific(v, unique=True, trim="0").partition("e")
    return "%se%d" % (m, int(e))


def runTimeSteps_Print(g, nsteps, out=print):
    """The solve loop of src/Main.hs:278-306, bounded to nsteps; prints the same five lines per step."""
    for step in range(nsteps):
        runSingleStep(g)
        out(str(g.n))
        out("step: %d" % step)
        out("timeLevel: %d" % g.currModTime)
        out("timeLevelAbsolute: %d" % g.currAbsTime)
        out("totalDensity: " + _hs_show_double(totalDensity(g)))
        out(" ")
    return g
