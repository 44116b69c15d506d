"""Several slabs behind one handle, driven by the library itself (cfd_multi_* in include/cfd_hip.h): the slab
decomposition, the halo exchange (RCCL send/recv or device copies) and the overlap of exchange and interior
compute all happen in C; this module only marshals arrays.

    MultiGraph(params, fields, active, n_slabs, devices=None, transport="copy")    every slab in this process
    MultiGraph.for_rank(params, fields, active, rank, world, comm_id)              one process per GPU (RCCL)
"""
import ctypes

import numpy as np

from . import _capi
from ._capi import CfdParams, check


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator: call on ONE rank, hand the bytes to every rank."""
    buf = ctypes.create_string_buffer(_capi.CFD_COMM_ID_BYTES)
    check(_capi.load().cfd_comm_unique_id(buf))
    return buf.raw


def rccl_selftest(device=0, nbytes=1 << 20):
    check(_capi.load().cfd_rccl_selftest(int(device), int(nbytes)))


class _Slab:
    """A local slab's single-slab handle, for timing and output slices (never stepped or destroyed here)."""

    def __init__(self, lib, h, x0, x1):
        self._lib, self._h, self.x0, self.x1 = lib, h, x0, x1

    def set_timing(self, enable):
        check(self._lib.cfd_set_timing(self._h, int(bool(enable))))

    def kernel_timing(self, kind, reset=True):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(self._lib.cfd_kernel_timing(self._h, int(kind), ctypes.byref(ms), ctypes.byref(n), int(bool(reset))))
        return ms.value, n.value

    def sweep_timing(self, reset=True):
        return self.kernel_timing(0, reset)

    def kernel_name(self):
        return self._lib.cfd_kernel_name(self._h).decode()

    @property
    def live_fields(self):
        n = ctypes.c_int()
        check(self._lib.cfd_live_fields(self._h, ctypes.byref(n)))
        return n.value


class MultiGraph:
    _TRANSPORT = {"copy": _capi.CFD_TRANSPORT_COPY, "rccl": _capi.CFD_TRANSPORT_RCCL}

    def __init__(self, params, fields, active, n_slabs, devices=None, transport="copy"):
        """params: the WHOLE grid (solver.make_params); fields / active as for AdjGraph."""
        self._lib = _capi.load()
        self._h = ctypes.c_void_p()
        f, act, p = self._marshal(params, fields, active)
        dev = None
        if devices is not None:
            dev = (ctypes.c_int32 * n_slabs)(*[int(d) for d in devices])
        check(self._lib.cfd_multi_create(ctypes.byref(p), int(n_slabs), dev, self._TRANSPORT[transport],
                                         _capi.ptr_array(f), act.ctypes.data, ctypes.byref(self._h)))

    @classmethod
    def for_rank(cls, params, fields, active, rank, world, comm_id):
        """params / fields / active: this rank's slab only.  comm_id: comm_unique_id() of one rank (None if world == 1)."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        self._h = ctypes.c_void_p()
        f, act, p = self._marshal(params, fields, active)
        check(self._lib.cfd_multi_create_rank(ctypes.byref(p), int(rank), int(world), comm_id, _capi.ptr_array(f),
                                              act.ctypes.data, ctypes.byref(self._h)))
        return self

    def _marshal(self, params, fields, active):
        self.params = dict(params)
        p = CfdParams(**self.params)
        self.n = p.nx * p.ny * p.nz
        f = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in fields]
        if len(f) != _capi.CFD_NPROP or any(x.size != self.n for x in f):
            raise ValueError("fields must be 7 arrays of nx*ny*nz values")
        act = np.ascontiguousarray(active, dtype=np.uint8).reshape(-1)
        if act.size != self.n:
            raise ValueError("active must have nx*ny*nz entries")
        return f, act, p

    def step(self, nsteps=1):
        check(self._lib.cfd_multi_step(self._h, int(nsteps)))
        return self

    def total_density(self):
        v = ctypes.c_double()
        check(self._lib.cfd_multi_total_density(self._h, ctypes.byref(v)))
        return v.value

    def read_fields(self, slot_rel=0):
        out = [np.empty(self.n, dtype=np.float64) for _ in range(_capi.CFD_NPROP)]
        check(self._lib.cfd_multi_read_fields(self._h, int(slot_rel), _capi.ptr_array(out)))
        return out

    def write_fields(self, fields, slot_rel=0):
        f = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in fields]
        if len(f) != _capi.CFD_NPROP or any(x.size != self.n for x in f):
            raise ValueError("fields must be 7 arrays of nx*ny*nz values")
        check(self._lib.cfd_multi_write_fields(self._h, int(slot_rel), _capi.ptr_array(f)))
        return self

    def set_time(self, curr_mod_time, curr_abs_time):
        check(self._lib.cfd_multi_set_time(self._h, int(curr_mod_time), int(curr_abs_time)))
        return self

    def checkpoint(self):
        """(currModTime, currAbsTime, [fields of slot_rel 0, 1, 2]) of the local slabs: everything a run needs to resume."""
        cmt, cat = self._time()
        return cmt, cat, [self.read_fields(r) for r in range(3)]

    def restore(self, state):
        cmt, cat, slots = state
        self.set_time(cmt, cat)
        for r in range(3):
            self.write_fields(slots[r], r)
        return self

    def _time(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        check(self._lib.cfd_multi_time(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    currModTime = property(lambda self: self._time()[0])
    currAbsTime = property(lambda self: self._time()[1])

    def info(self):
        """(local slabs; 2 = the whole step is one pass with one exchange, 1 = the step's tail is one pass, 0 = three
        sweeps with an exchange each; exchanges beside the interior planes)"""
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(self._lib.cfd_multi_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, bool(c.value)

    def set_exchange_timing(self, enable):
        check(self._lib.cfd_multi_set_exchange_timing(self._h, 1 if enable else 0))

    def exchange_timing(self, reset=False):
        """(total milliseconds, number) of the first local slab's ghost-plane exchanges while the timing was on;
        waits for the queued work first."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(self._lib.cfd_multi_exchange_timing(self._h, ctypes.byref(ms), ctypes.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def slab(self, i):
        h, x0, x1 = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        check(self._lib.cfd_multi_slab(self._h, int(i), ctypes.byref(h), ctypes.byref(x0), ctypes.byref(x1)))
        return _Slab(self._lib, h, x0.value, x1.value)

    def sync(self):
        check(self._lib.cfd_multi_sync(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.cfd_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
