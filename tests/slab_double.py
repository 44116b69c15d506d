"""CPU stand-in for the GPU backend of cfd_amd.slab, built on Oracle B, so that the slab
orchestration (who sends which plane to whom, when, and the ghost flags) can run under gloo.

The oracle runs on the slab extended by TWO ghost planes on every side that has a neighbour.  A
sweep also "updates" the ghost planes: the outermost one with a wrongly clamped stencil, the one
next to the slab exactly, provided both were exchanged and its active mask is the neighbour's
(set_ghost_active).  An owned plane's stencil reaches only centre values and in-plane
reconstructions of the plane beside it, so
  * after a one-plane exchange one sweep leaves the owned planes exact,
  * after a two-plane exchange two sweeps in a row do (sweep_pair_and_filter),
and every ghost plane is overwritten by the next exchange before it is read again.
"""
import numpy as np
import torch

from helpers import oracle_params
from oracle import OracleB

GHOST = 2
LIVE = (0, 1, 2, 3, 4, 6)       # propIndex of the oracle's six live fields
OUTPUT = ("Speed", "Vorticity", "Pressure", "U", "V", "W", "Density", "Temperature", "Mew", "Dye")


class OracleSlabBackend:
    def __init__(self, params, fields, active):
        self.gl, self.gh = params["ghost_lo"], params["ghost_hi"]
        self.flags = int(params.get("flags", 0))        # what the slab driver asked of the library
        self.nx = params["nx"]
        self.plane = params["ny"] * params["nz"]
        ext = dict(oracle_params(params))
        self.lo, self.hi = GHOST * self.gl, GHOST * self.gh     # ghost planes below / above the owned ones
        ext["nx"] = self.nx + self.lo + self.hi
        pad = lambda a, fill: np.concatenate([np.full(self.lo * self.plane, fill, a.dtype), a,
                                              np.full(self.hi * self.plane, fill, a.dtype)])
        # ghost planes start as 1.0 (any finite non-zero value): they are filled before the first sweep
        self.ob = OracleB(ext, [pad(np.asarray(f, np.float64), 1.0) for f in fields],
                          pad(np.asarray(active, np.uint8), 0), pow_mode=1)
        act = np.asarray(active, np.uint8).reshape(-1)
        self._first_mask, self._last_mask = act[:self.plane].copy(), act[-self.plane:].copy()
        self._recv = {d: [torch.zeros(d * 6 * self.plane, dtype=torch.float64) for _ in range(2)] for d in (1, 2)}
        self._masks = not (self.gl or self.gh)

    def _plane(self, q, x):
        v = self.ob.slot_view(0, q)
        return v[x * self.plane:(x + 1) * self.plane]

    def _pack(self, x0, depth):
        # plane-major, then field: the order in which the device layout keeps a run of planes
        return torch.from_numpy(np.concatenate([self._plane(q, x) for x in range(x0, x0 + depth) for q in range(6)]))

    def _unpack(self, buf, x0, depth):
        b = buf.numpy()
        for i in range(depth):
            for q in range(6):
                k = i * 6 + q
                self._plane(q, x0 + i)[:] = b[k * self.plane:(k + 1) * self.plane]

    def halo_tensors(self, depth=1):
        recv = self._recv[depth]
        return (self._pack(self.lo, depth), self._pack(self.lo + self.nx - depth, depth),
                (recv[0] if self.gl else None), (recv[1] if self.gh else None))

    def halo_commit(self, depth=1):
        if self.gl:
            self._unpack(self._recv[depth][0], self.lo - depth, depth)
        if self.gh:
            self._unpack(self._recv[depth][1], self.lo + self.nx, depth)

    # the masks of the neighbours' planes next to this slab (cfd_set_ghost_active)
    def boundary_active(self):
        return self._first_mask, self._last_mask

    def mask_tensor(self, plane=None):
        if plane is None:
            return torch.zeros(self.plane, dtype=torch.uint8)
        return torch.from_numpy(np.ascontiguousarray(plane).copy())

    def set_ghost_active(self, lo_plane, hi_plane):
        a = self.ob.active_view()
        if self.gl:
            a[(self.lo - 1) * self.plane:self.lo * self.plane] = lo_plane
        if self.gh:
            a[(self.lo + self.nx) * self.plane:(self.lo + self.nx + 1) * self.plane] = hi_plane
        self._masks = True

    def pair_available(self):
        return self._masks and self.nx >= 2

    def sweep_pair_and_filter(self):
        if not self.pair_available():
            raise RuntimeError("sweep_pair_and_filter: ghost masks missing or slab too thin")
        self.ob.sweep(False)
        self.ob.sweep(False)
        self.ob.time_filter()

    def scalar_tensor(self, v):
        return torch.tensor([v], dtype=torch.float64)

    def sweep(self, step_time):
        self.ob.sweep(step_time)

    def time_filter(self):
        self.ob.time_filter()

    def sweep_and_filter(self):
        self.ob.sweep(False)
        self.ob.time_filter()

    # split sweep: the oracle has no partial sweep, so the whole sweep runs at the end -- by which
    # time the orchestration under test must have delivered the ghost planes
    def sweep_begin(self, step_time, with_filter=False, edge=1):
        self._open = (step_time, with_filter)

    def sweep_end(self):
        step_time, with_filter = self._open
        self.ob.sweep(step_time)
        if with_filter:
            self.ob.time_filter()

    def _own(self, a):
        return a[self.lo * self.plane:(self.lo + self.nx) * self.plane]

    def total_density(self):
        acc = 0.0
        for v in self._own(self.ob.read_fields(1)[3]):
            acc += v
        return acc

    def read_fields(self, slot_rel=0):
        return [self._own(f).copy() for f in self.ob.read_fields(slot_rel)]

    def property_slice(self, prp, z_level=None):
        k = OUTPUT.index(prp) if isinstance(prp, str) else int(prp)
        nz = self.ob.p.nz
        rows = self.ob.property_slice(k, (nz - 1) // 2 if z_level is None else z_level)
        return rows[self.lo:self.lo + self.nx].copy()

    def sync(self):
        pass

    def close(self):
        self.ob.close()
