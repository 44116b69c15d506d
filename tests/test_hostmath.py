"""The kernels' per-cell arithmetic (cfd_amd/csrc/sweep_math.h) compiled for the host and checked
bit-for-bit against Oracle B: pins the operation order of the HIP code without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import bits_equal, field_errors, oracle_params, small_case
from oracle import OracleB

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm():
    d = os.path.join(HERE, "hostmath")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(os.path.join(d, "libhostmath.so"))


def _ptrs(arrs):
    a = (ctypes.c_void_p * len(arrs))()
    for i, x in enumerate(arrs):
        a[i] = x.ctypes.data
    return a


@pytest.mark.parametrize("shape,seed,zero_v", [((12, 9), 1, False), ((31, 17), 2, True), ((64, 40), 3, False),
                                                ((1, 1), 4, False), ((2, 5), 5, False)])
def test_sweep2d_math_bit_exact(hm, shape, seed, zero_v):
    nx, ny = shape
    params, fields, active = small_case(nx, ny, 1, seed, zero_v)
    p = oracle_params(params)
    b = OracleB(p, fields, active, pow_mode=1)
    cur = [fields[i].copy() for i in (0, 1, 3, 4, 6)]
    d = ctypes.c_double
    for sweep in range(4):
        b.sweep(sweep == 0)
        out = [np.empty(nx * ny) for _ in range(5)]
        hm.hm_sweep2d(nx, ny, d(p["dx"]), d(p["dy"]), d(p["dz"]), d(p["dt"]), d(p["lambda_"]), d(p["suth"]),
                      d(p["R"]), d(p["cv"]), d(p["k"]), d(p["dye_f"]), _ptrs(cur),
                      active.ctypes.data_as(ctypes.c_void_p), _ptrs(out))
        fb = b.read_fields(0)
        for j, i in enumerate((0, 1, 3, 4, 6)):
            assert bits_equal(out[j], fb[i]), (sweep, i)
        assert np.all(fb[2] == 0)
        cur = out


@pytest.mark.parametrize("shape,seed,zero_v", [((7, 6, 5), 1, False), ((3, 3, 3), 2, True), ((12, 9, 1), 3, False),
                                                ((4, 1, 6), 4, False), ((1, 1, 1), 5, False), ((9, 8, 2), 6, True)])
def test_sweep3d_math_bit_exact(hm, shape, seed, zero_v):
    nx, ny, nz = shape
    params, fields, active = small_case(nx, ny, nz, seed, zero_v)
    p = oracle_params(params)
    b = OracleB(p, fields, active, pow_mode=1)
    live = (0, 1, 2, 3, 4, 6)
    cur = [fields[i].copy() for i in live]
    d = ctypes.c_double
    for sweep in range(4):
        b.sweep(sweep == 0)
        out = [np.empty(nx * ny * nz) for _ in range(6)]
        hm.hm_sweep3d(nx, ny, nz, d(p["dx"]), d(p["dy"]), d(p["dz"]), d(p["dt"]), d(p["lambda_"]), d(p["suth"]),
                      d(p["R"]), d(p["cv"]), d(p["k"]), d(p["dye_f"]), _ptrs(cur),
                      active.ctypes.data_as(ctypes.c_void_p), _ptrs(out))
        fb = b.read_fields(0)
        for j, i in enumerate(live):
            assert bits_equal(out[j], fb[i]), (sweep, i, float(np.max(np.abs(out[j] - fb[i]))))
        cur = out


@pytest.mark.parametrize("scale", [1e-290, 1e-300, 1e-305, 1e-310, 1e-318])
@pytest.mark.parametrize("nz", [1, 4])
def test_strict_math_bit_exact_with_subnormal_dye(hm, scale, nz):
    """Dye decays towards zero away from its source; its diffusion quotients then fall into the
    subnormal range, where the strict arithmetic must still match the oracle bit for bit."""
    nx, ny = 14, 11
    params, fields, active = small_case(nx, ny, nz, 3)
    fields = [f.copy() for f in fields]
    fields[6] = np.abs(fields[6]) * scale + scale
    p = oracle_params(params)
    b = OracleB(p, fields, active, pow_mode=1)
    live = (0, 1, 3, 4, 6) if nz == 1 else (0, 1, 2, 3, 4, 6)
    if nz == 1:
        fields[2][:] = 0.0
        b.close()
        b = OracleB(p, fields, active, pow_mode=1)
    cur = [fields[i].copy() for i in live]
    d = ctypes.c_double
    for sweep in range(3):
        b.sweep(sweep == 0)
        out = [np.empty(nx * ny * nz) for _ in live]
        if nz == 1:
            hm.hm_sweep2d(nx, ny, d(p["dx"]), d(p["dy"]), d(p["dz"]), d(p["dt"]), d(p["lambda_"]), d(p["suth"]),
                          d(p["R"]), d(p["cv"]), d(p["k"]), d(p["dye_f"]), _ptrs(cur),
                          active.ctypes.data_as(ctypes.c_void_p), _ptrs(out))
        else:
            hm.hm_sweep3d(nx, ny, nz, d(p["dx"]), d(p["dy"]), d(p["dz"]), d(p["dt"]), d(p["lambda_"]), d(p["suth"]),
                          d(p["R"]), d(p["cv"]), d(p["k"]), d(p["dye_f"]), _ptrs(cur),
                          active.ctypes.data_as(ctypes.c_void_p), _ptrs(out))
        fb = b.read_fields(0)
        for j, i in enumerate(live):
            assert bits_equal(out[j], fb[i]), (sweep, i)
        cur = out


# Re-associated ("fast") arithmetic: not bit-identical by construction.  Bound per sweep, starting
# each sweep from the oracle's own state: 1e-13 of the field scale (observed: below 1e-14).
FAST_TOL = 1e-13


@pytest.mark.parametrize("shape,seed,zero_v", [((12, 9), 1, False), ((31, 17), 2, True), ((64, 40), 3, False),
                                                ((1, 1), 4, False), ((2, 5), 5, False)])
@pytest.mark.parametrize("entry", ["hm_sweep2d_fast", "hm_sweep2d_stream"])
def test_sweep2d_fast_math_close(hm, shape, seed, zero_v, entry):
    nx, ny = shape
    params, fields, active = small_case(nx, ny, 1, seed, zero_v)
    p = oracle_params(params)
    b = OracleB(p, fields, active, pow_mode=1)
    live = (0, 1, 3, 4, 6)
    d = ctypes.c_double
    for sweep in range(4):
        cur = [b.read_fields(0)[i].copy() for i in live]
        b.sweep(sweep == 0)
        out = [np.empty(nx * ny) for _ in range(5)]
        getattr(hm, entry)(nx, ny, d(p["dx"]), d(p["dy"]), d(p["dz"]), d(p["dt"]), d(p["lambda_"]), d(p["suth"]),
                           d(p["R"]), d(p["cv"]), d(p["k"]), d(p["dye_f"]), _ptrs(cur),
                           active.ctypes.data_as(ctypes.c_void_p), _ptrs(out))
        fb = b.read_fields(0)
        full = [out[0], out[1], fb[2], out[2], out[3], fb[5], out[4]]
        assert max(field_errors(full, fb)) <= FAST_TOL, (sweep, field_errors(full, fb))


@pytest.mark.parametrize("shape,seed,zero_v", [((7, 6, 5), 1, False), ((3, 3, 3), 2, True), ((12, 9, 1), 3, False),
                                                ((4, 1, 6), 4, False), ((1, 1, 1), 5, False), ((9, 8, 2), 6, True)])
@pytest.mark.parametrize("entry", ["hm_sweep3d_fast", "hm_sweep3d_stream"])
def test_sweep3d_fast_math_close(hm, shape, seed, zero_v, entry):
    nx, ny, nz = shape
    params, fields, active = small_case(nx, ny, nz, seed, zero_v)
    p = oracle_params(params)
    b = OracleB(p, fields, active, pow_mode=1)
    live = (0, 1, 2, 3, 4, 6)
    d = ctypes.c_double
    for sweep in range(4):
        cur = [b.read_fields(0)[i].copy() for i in live]
        b.sweep(sweep == 0)
        out = [np.empty(nx * ny * nz) for _ in range(6)]
        getattr(hm, entry)(nx, ny, nz, d(p["dx"]), d(p["dy"]), d(p["dz"]), d(p["dt"]), d(p["lambda_"]),
                           d(p["suth"]), d(p["R"]), d(p["cv"]), d(p["k"]), d(p["dye_f"]), _ptrs(cur),
                           active.ctypes.data_as(ctypes.c_void_p), _ptrs(out))
        fb = b.read_fields(0)
        full = [out[0], out[1], out[2], out[3], out[4], fb[5], out[5]]
        assert max(field_errors(full, fb)) <= FAST_TOL, (sweep, field_errors(full, fb))


def test_half_superbee_special_values(hm):
    """The kernels' branch-free half-limiter against the literal restatement, including zeros,
    denormals, huge values and mixed signs (the sign of a zero result is not significant)."""
    from oracle import oracle_a as A
    hm.hm_half_superbee.restype = ctypes.c_double
    hm.hm_half_superbee.argtypes = [ctypes.c_double, ctypes.c_double]
    vals = [0.0, -0.0, 5e-324, -5e-324, 1e-310, 1e-300, 1.0, -1.0, 2.0, 3.0, -3.0, 0.5, 1e300, -1e300,
            1.7976931348623157e308, 2.5, -2.5, 1e-17]
    for a in vals:
        for b in vals:
            want = 0.5 * A.superbee(a, b)
            got = hm.hm_half_superbee(a, b)
            assert got == want, (a, b, got, want)
    rng = np.random.default_rng(9)
    for _ in range(3000):
        a, b = rng.normal(size=2) * 10.0 ** rng.integers(-12, 12)
        assert hm.hm_half_superbee(a, b) == 0.5 * A.superbee(a, b)


def test_div_by_constant_is_correctly_rounded(hm):
    """div_by (four multiply-adds with the reciprocal held as a correctly rounded high part and a low part)
    against the IEEE division on 2e6 random numerators for each interval constant of several grids, for
    divisors at the edges of a binade (significand all ones, one ulp above a power of two, powers of two)
    and for random divisors: zero mismatches."""
    hm.hm_div_by_mismatches.restype = ctypes.c_long
    hm.hm_div_by_mismatches.argtypes = [ctypes.c_double, ctypes.c_long, ctypes.c_ulonglong]
    from cfd_amd import cases
    hs = set()
    for dx, dy, dz in [(cases.REF_DX, cases.REF_DY, 1.0), (cases.REF_DX, cases.REF_DY, cases.REF_DX),
                       (0.1, 0.3, 0.7), (1.0 / 3.0, 1e-3, 123.456)]:
        for ln in (dx, dy, dz):
            hs.add((0.0 + ln / 2.0) + ln / 2.0)
            hs.add(2.0 * (((0.0 + ln / 3.0) + ln / 3.0) + ln / 3.0))
    for e in (-9, 0, 3):
        hs.update([np.ldexp(1.0, e), np.nextafter(np.ldexp(1.0, e), 0.0), np.nextafter(np.ldexp(1.0, e), 4.0),
                   np.ldexp(1.5, e), np.nextafter(np.ldexp(1.5, e), 0.0)])
    rng = np.random.default_rng(77)
    hs.update(np.ldexp(rng.uniform(1.0, 2.0, 24), rng.integers(-12, 4, 24)).tolist())
    for i, h in enumerate(sorted(float(h) for h in hs)):
        assert hm.hm_div_by_mismatches(h, 2_000_000, 12345 + i) == 0, h


def test_host_arithmetic_under_sanitizers():
    """Every host entry point of the harness on small ragged grids (1x1 up to 17x4, 7x2x6), built with
    AddressSanitizer and UndefinedBehaviorSanitizer: `make -C tests/hostmath sanitize` must build and exit 0."""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(HERE, "hostmath"), "-B", "sanitize"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "sanitize: ok" in r.stdout
