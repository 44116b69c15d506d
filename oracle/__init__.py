"""CPU oracles for the field update.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by cfd_amd/.

oracle_a  -- literal restatement of the reference's term algebra (Python, tiny grids)
oracle_b  -- closed-form restatement (C, liboracle_b.so), checked bit-for-bit against oracle_a
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class ObParams(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32),
                ("dx", ctypes.c_double), ("dy", ctypes.c_double), ("dz", ctypes.c_double),
                ("dt", ctypes.c_double),
                ("lambda_", ctypes.c_double), ("suth", ctypes.c_double), ("R", ctypes.c_double),
                ("cv", ctypes.c_double), ("k", ctypes.c_double), ("dye_f", ctypes.c_double),
                ("pow_mode", ctypes.c_int32), ("nthreads", ctypes.c_int32)]


def build(force=False):
    """Compile liboracle_b.so with the committed Makefile (gcc only; no reference sources involved)."""
    so = os.path.join(_HERE, "liboracle_b.so")
    src = os.path.join(_HERE, "oracle_b.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_b.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_b.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        PP = ctypes.POINTER(ctypes.c_void_p)
        L.ob_create.restype = ctypes.c_void_p
        L.ob_create.argtypes = [ctypes.POINTER(ObParams), PP, ctypes.c_void_p]
        L.ob_destroy.argtypes = [ctypes.c_void_p]
        L.ob_sweep.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ob_time_filter.argtypes = [ctypes.c_void_p]
        L.ob_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ob_total_density.restype = ctypes.c_double
        L.ob_total_density.argtypes = [ctypes.c_void_p]
        L.ob_read_fields.argtypes = [ctypes.c_void_p, ctypes.c_int, PP]
        L.ob_time.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.ob_slot_ptr.restype = ctypes.POINTER(ctypes.c_double)
        L.ob_slot_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.ob_property.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.ob_active_ptr.restype = ctypes.POINTER(ctypes.c_uint8)
        L.ob_active_ptr.argtypes = [ctypes.c_void_p]
        L.ob_superbee.restype = ctypes.c_double
        L.ob_superbee.argtypes = [ctypes.c_double, ctypes.c_double]
        L.ob_mu.restype = ctypes.c_double
        L.ob_mu.argtypes = [ctypes.POINTER(ObParams), ctypes.c_double]
        _LIB = L
    return _LIB


def _ptr_array(arrs):
    a = (ctypes.c_void_p * len(arrs))()
    for i, x in enumerate(arrs):
        a[i] = x.ctypes.data
    return a


class OracleB:
    """Thin handle on liboracle_b.so with the same verbs as the product library."""

    def __init__(self, params, fields7, active, pow_mode=0, nthreads=1):
        """params: dict with nx,ny,nz,dx,dy,dz,dt,lambda_,suth,R,cv,k,dye_f."""
        self.L = lib()
        self.p = ObParams(pow_mode=pow_mode, nthreads=nthreads, **params)
        self.n = params["nx"] * params["ny"] * params["nz"]
        f = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in fields7]
        assert len(f) == 7 and all(x.size == self.n for x in f)
        act = np.ascontiguousarray(active, dtype=np.uint8).reshape(-1)
        assert act.size == self.n
        self.h = self.L.ob_create(ctypes.byref(self.p), _ptr_array(f), act.ctypes.data)

    def sweep(self, step_time):
        self.L.ob_sweep(self.h, int(bool(step_time)))

    def time_filter(self):
        self.L.ob_time_filter(self.h)

    def step(self, nsteps=1):
        self.L.ob_step(self.h, nsteps)

    def total_density(self):
        return self.L.ob_total_density(self.h)

    def read_fields(self, slot_rel=0):
        out = [np.empty(self.n, dtype=np.float64) for _ in range(7)]
        self.L.ob_read_fields(self.h, slot_rel, _ptr_array(out))
        return out

    def property_field(self, prop, slot_rel=0):
        """Output property (reference Property enum: 0 Speed 1 Vorticity 2 Pressure 3 U 4 V 5 W 6 Density
        7 Temperature 8 Mew 9 Dye) of every cell of slot cmt-slot_rel, node order."""
        out = np.empty(self.n, dtype=np.float64)
        self.L.ob_property(self.h, int(prop), int(slot_rel), out.ctypes.data)
        return out

    def property_slice(self, prop, z_level, slot_rel=0):
        """adjGraphToGrid (src/Main.hs:339-346): the z = z_level slice as an (nx, ny) array."""
        p = self.p
        return self.property_field(prop, slot_rel).reshape(p.nx, p.ny, p.nz)[:, :, z_level].copy()

    def slot_view(self, slot_rel, q):
        """Writable numpy view of live field q (U V W Density Temperature Dye) of slot cmt-slot_rel."""
        return np.ctypeslib.as_array(self.L.ob_slot_ptr(self.h, slot_rel, q), shape=(self.n,))

    def active_view(self):
        """Writable numpy view of the active mask."""
        return np.ctypeslib.as_array(self.L.ob_active_ptr(self.h), shape=(self.n,))

    def time(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        self.L.ob_time(self.h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def close(self):
        if self.h:
            self.L.ob_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
