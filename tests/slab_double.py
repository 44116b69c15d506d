"""CPU stand-in for the GPU backend of cfd_amd.slab, built on Oracle B, so that the slab
orchestration (who sends which plane to whom, when, and the ghost flags) can run under gloo.

The oracle runs on the slab extended by its ghost planes.  A sweep also "updates" the ghost planes
with a wrongly clamped stencil, but an owned plane's stencil reaches only centre values and
in-plane reconstructions of a ghost plane, so owned planes are exact, and every ghost plane is
overwritten by the exchange before it is read again.
"""
import numpy as np
import torch

from helpers import oracle_params
from oracle import OracleB

LIVE = (0, 1, 2, 3, 4, 6)       # propIndex of the oracle's six live fields
OUTPUT = ("Speed", "Vorticity", "Pressure", "U", "V", "W", "Density", "Temperature", "Mew", "Dye")


class OracleSlabBackend:
    def __init__(self, params, fields, active):
        self.gl, self.gh = params["ghost_lo"], params["ghost_hi"]
        self.nx = params["nx"]
        self.plane = params["ny"] * params["nz"]
        ext = dict(oracle_params(params))
        ext["nx"] = self.nx + self.gl + self.gh
        pad = lambda a, fill: np.concatenate([np.full(self.gl * self.plane, fill, a.dtype), a,
                                              np.full(self.gh * self.plane, fill, a.dtype)])
        # ghost planes start as 1.0 (any finite non-zero value): they are filled before the first sweep
        self.ob = OracleB(ext, [pad(np.asarray(f, np.float64), 1.0) for f in fields],
                          pad(np.asarray(active, np.uint8), 0), pow_mode=1)
        self._recv = [torch.zeros(6 * self.plane, dtype=torch.float64) for _ in range(2)]

    def _plane(self, q, x):
        v = self.ob.slot_view(0, q)
        return v[x * self.plane:(x + 1) * self.plane]

    def halo_tensors(self):
        first, last = self.gl, self.gl + self.nx - 1
        send_lo = torch.from_numpy(np.concatenate([self._plane(q, first) for q in range(6)]))
        send_hi = torch.from_numpy(np.concatenate([self._plane(q, last) for q in range(6)]))
        return send_lo, send_hi, (self._recv[0] if self.gl else None), (self._recv[1] if self.gh else None)

    def halo_commit(self):
        if self.gl:
            for q in range(6):
                self._plane(q, 0)[:] = self._recv[0].numpy()[q * self.plane:(q + 1) * self.plane]
        if self.gh:
            for q in range(6):
                self._plane(q, self.gl + self.nx)[:] = self._recv[1].numpy()[q * self.plane:(q + 1) * self.plane]

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
    def sweep_begin(self, step_time, with_filter=False):
        self._open = (step_time, with_filter)

    def sweep_end(self):
        step_time, with_filter = self._open
        self.ob.sweep(step_time)
        if with_filter:
            self.ob.time_filter()

    def _own(self, a):
        return a[self.gl * self.plane:(self.gl + self.nx) * self.plane]

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
        return rows[self.gl:self.gl + self.nx].copy()

    def sync(self):
        pass

    def close(self):
        self.ob.close()
