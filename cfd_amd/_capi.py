"""ctypes binding of libcfdhip.so (the C ABI in include/cfd_hip.h).

The library is the only compute path: if it is missing, or no GPU is present, every call
fails loudly.  There is no CPU fallback in this package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CFD_LIB: another build of the same library (A/B timing of kernel changes)
LIB_PATH = os.environ.get("CFD_LIB") or os.path.join(_HERE, "libcfdhip.so")

CFD_NPROP = 7
CFD_OK, CFD_E_ARG, CFD_E_DEVICE, CFD_E_MEMORY, CFD_E_STATE = 0, -1, -2, -3, -4
CFD_FLAG_DEFAULT, CFD_FLAG_SIMPLE_KERNEL, CFD_FLAG_FORCE_3D, CFD_FLAG_FAST_MATH = 0, 1, 2, 4
CFD_TRANSPORT_COPY, CFD_TRANSPORT_RCCL = 0, 1
CFD_COMM_ID_BYTES = 128

# every symbol include/cfd_hip.h declares
EXPORTS = [
    "cfd_default_params", "cfd_create", "cfd_destroy", "cfd_step", "cfd_sweep", "cfd_sweep_and_filter", "cfd_sweep_begin", "cfd_sweep_end", "cfd_sweep_begin_depth", "cfd_sweep_end_on", "cfd_sweep_pair_and_filter", "cfd_time_filter",
    "cfd_total_density", "cfd_total_density_noalloc", "cfd_read_fields", "cfd_write_fields", "cfd_set_time", "cfd_property_slice", "cfd_time", "cfd_last_error", "cfd_halo_planes", "cfd_halo_planes_depth", "cfd_set_ghost_active", "cfd_pair_available",
    "cfd_set_ghost_active_depth", "cfd_one_pass_step_available", "cfd_step_one_pass", "cfd_step_one_pass_begin", "cfd_step_one_pass_end_on",
    "cfd_multi_create", "cfd_multi_create_rank", "cfd_comm_unique_id", "cfd_multi_destroy", "cfd_multi_step", "cfd_multi_total_density",
    "cfd_multi_read_fields", "cfd_multi_write_fields", "cfd_multi_set_time", "cfd_multi_time", "cfd_multi_sync", "cfd_multi_info", "cfd_multi_slab", "cfd_multi_set_exchange_timing", "cfd_multi_exchange_timing", "cfd_rccl_selftest",
    "cfd_stream", "cfd_set_stream", "cfd_kernel_name", "cfd_sync", "cfd_set_timing", "cfd_sweep_timing", "cfd_kernel_timing", "cfd_live_fields",
]


class CfdParams(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32),
                ("dx", ctypes.c_double), ("dy", ctypes.c_double), ("dz", ctypes.c_double),
                ("dt", ctypes.c_double),
                ("lambda_", ctypes.c_double), ("suth", ctypes.c_double), ("R", ctypes.c_double),
                ("cv", ctypes.c_double), ("k", ctypes.c_double), ("dye_f", ctypes.c_double),
                ("device", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("ghost_lo", ctypes.c_int32), ("ghost_hi", ctypes.c_int32)]


class CfdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcfdhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libcfdhip.so and declare its signatures.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # A process must hold ONE HIP runtime.  PyTorch ships its own copy of libamdhip64 under the same
    # soname as the system one this library links against; whichever is loaded first serves both.
    # If the system copy came first, torch would later find no GPU -- so when torch is installed,
    # let it load its runtime before this library binds to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libcfdhip.so is not built (%s); run `make -C cfd_amd/csrc` or __graft_entry__.build(). "
            "There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    H = ctypes.c_void_p
    PP = ctypes.POINTER(ctypes.c_void_p)
    L.cfd_default_params.argtypes = [ctypes.POINTER(CfdParams)]
    L.cfd_create.argtypes = [ctypes.POINTER(CfdParams), PP, ctypes.c_void_p, ctypes.POINTER(H)]
    L.cfd_destroy.argtypes = [H]
    L.cfd_destroy.restype = None
    L.cfd_step.argtypes = [H, ctypes.c_int]
    L.cfd_sweep.argtypes = [H, ctypes.c_int]
    L.cfd_time_filter.argtypes = [H]
    L.cfd_sweep_and_filter.argtypes = [H]
    L.cfd_sweep_begin.argtypes = [H, ctypes.c_int, ctypes.c_int]
    L.cfd_sweep_end.argtypes = [H]
    L.cfd_sweep_begin_depth.argtypes = [H, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.cfd_sweep_end_on.argtypes = [H, ctypes.c_void_p]
    L.cfd_sweep_pair_and_filter.argtypes = [H]
    L.cfd_total_density.argtypes = [H, ctypes.POINTER(ctypes.c_double)]
    L.cfd_total_density_noalloc.argtypes = [H, ctypes.c_void_p]
    L.cfd_read_fields.argtypes = [H, ctypes.c_int, PP]
    L.cfd_write_fields.argtypes = [H, ctypes.c_int, PP]
    L.cfd_set_time.argtypes = [H, ctypes.c_int, ctypes.c_int]
    L.cfd_property_slice.argtypes = [H, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.cfd_time.argtypes = [H, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.cfd_last_error.restype = ctypes.c_char_p
    L.cfd_halo_planes.argtypes = [H, PP, PP, PP, PP, ctypes.POINTER(ctypes.c_size_t)]
    L.cfd_halo_planes_depth.argtypes = [H, ctypes.c_int, PP, PP, PP, PP, ctypes.POINTER(ctypes.c_size_t)]
    L.cfd_set_ghost_active.argtypes = [H, ctypes.c_void_p, ctypes.c_void_p]
    L.cfd_pair_available.argtypes = [H, ctypes.POINTER(ctypes.c_int)]
    L.cfd_set_ghost_active_depth.argtypes = [H, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.cfd_one_pass_step_available.argtypes = [H, ctypes.POINTER(ctypes.c_int)]
    L.cfd_step_one_pass.argtypes = [H]
    L.cfd_step_one_pass_begin.argtypes = [H]
    L.cfd_step_one_pass_end_on.argtypes = [H, ctypes.c_void_p]
    L.cfd_stream.argtypes = [H, PP]
    L.cfd_sync.argtypes = [H]
    L.cfd_set_stream.argtypes = [H, ctypes.c_void_p]
    L.cfd_kernel_name.argtypes = [H]
    L.cfd_kernel_name.restype = ctypes.c_char_p
    L.cfd_set_timing.argtypes = [H, ctypes.c_int]
    L.cfd_sweep_timing.argtypes = [H, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    L.cfd_kernel_timing.argtypes = [H, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                    ctypes.c_int]
    L.cfd_live_fields.argtypes = [H, ctypes.POINTER(ctypes.c_int)]
    PI = ctypes.POINTER(ctypes.c_int)
    L.cfd_multi_create.argtypes = [ctypes.POINTER(CfdParams), ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, PP,
                                   ctypes.c_void_p, ctypes.POINTER(H)]
    L.cfd_comm_unique_id.argtypes = [ctypes.c_void_p]
    L.cfd_multi_create_rank.argtypes = [ctypes.POINTER(CfdParams), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, PP,
                                        ctypes.c_void_p, ctypes.POINTER(H)]
    L.cfd_multi_destroy.argtypes = [H]
    L.cfd_multi_destroy.restype = None
    L.cfd_multi_step.argtypes = [H, ctypes.c_int]
    L.cfd_multi_total_density.argtypes = [H, ctypes.POINTER(ctypes.c_double)]
    L.cfd_multi_read_fields.argtypes = [H, ctypes.c_int, PP]
    L.cfd_multi_time.argtypes = [H, PI, PI]
    L.cfd_multi_write_fields.argtypes = [H, ctypes.c_int, PP]
    L.cfd_multi_set_time.argtypes = [H, ctypes.c_int, ctypes.c_int]
    L.cfd_multi_sync.argtypes = [H]
    L.cfd_multi_info.argtypes = [H, PI, PI, PI]
    L.cfd_multi_set_exchange_timing.argtypes = [H, ctypes.c_int]
    L.cfd_multi_exchange_timing.argtypes = [H, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    L.cfd_multi_slab.argtypes = [H, ctypes.c_int, ctypes.POINTER(H), PI, PI]
    L.cfd_rccl_selftest.argtypes = [ctypes.c_int, ctypes.c_size_t]
    for name in EXPORTS:
        if name not in ("cfd_destroy", "cfd_multi_destroy", "cfd_last_error", "cfd_kernel_name"):
            getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def check(rc):
    if rc != CFD_OK:
        raise CfdError(rc, load().cfd_last_error().decode("utf-8", "replace"))


def ptr_array(arrs):
    a = (ctypes.c_void_p * len(arrs))()
    for i, x in enumerate(arrs):
        a[i] = x.ctypes.data
    return a
