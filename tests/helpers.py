"""Shared test scaffolding: seeded small cases, oracle construction, comparison metrics."""
import numpy as np

from cfd_amd import cases
from cfd_amd.solver import make_params

PROPS = ("U", "V", "W", "Density", "Temperature", "Mew", "Dye")


def oracle_params(params):
    """cfd_amd parameter dict -> OracleB parameter dict."""
    keys = ("nx", "ny", "nz", "dx", "dy", "dz", "dt", "lambda_", "suth", "R", "cv", "k", "dye_f")
    return {k: params[k] for k in keys}


def small_case(nx, ny, nz=1, seed=0, zero_v=False, amp=1.0):
    """Perturbed small grid with an inflow column and a 2x2(xnz) obstacle; every limiter branch,
    the domain-edge clamps and (with zero_v) the NaN->1 coefficient rule are exercised."""
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    u = lambda: rng.uniform(-1, 1, n)
    U = 10 + 0.5 * amp * u()
    V = 0.3 * amp * u()
    W = 0.2 * amp * u() if nz > 1 else np.zeros(n)
    if zero_v:
        V[:] = 0.0
    rho = 0.3 * (1 + 1e-2 * amp * u())
    T = 223 * (1 + 1e-2 * amp * u())
    mew = np.full(n, 0.1)
    dye = rng.uniform(0, 1, n)
    # a few exactly equal neighbours so that the limiter's Zero branch is taken
    T.reshape(nx, ny, nz)[nx // 2, :, :] = T.reshape(nx, ny, nz)[nx // 2 - 1, :, :]
    dye.reshape(nx, ny, nz)[:, ny // 2, :] = 0.0
    act = np.ones((nx, ny, nz), np.uint8)
    act[0] = 0
    ox, oy = nx // 3, ny // 3
    act[ox:ox + 2, oy:oy + 2, :] = 0
    for f in (U, V, W):
        f.reshape(nx, ny, nz)[ox:ox + 2, oy:oy + 2, :] = 0.0
    dz = cases.REF_DZ if nz == 1 else cases.REF_DX
    params = make_params(nx, ny, nz, cases.REF_DX, cases.REF_DY, dz)
    return params, [U, V, W, rho, T, mew, dye], act.reshape(n)


def bits_equal(a, b):
    """Bitwise equality, except that +0 and -0 compare equal."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | ((a == 0) & (b == 0))))


def rel_err(a, b):
    """max |a-b| / max |b| : field-scale relative error."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.max(np.abs(b))
    d = np.max(np.abs(a - b))
    return 0.0 if d == 0 else float(d / scale) if scale > 0 else float("inf")


def field_errors(got, want):
    """Per-property error of two 7-field lists, each relative to its field's scale: max|want| for
    Density, Temperature, Dye; the largest velocity component magnitude for U, V, W (they are
    components of one vector and carry the same absolute rounding noise)."""
    vs = max(float(np.max(np.abs(want[k]))) for k in (0, 1, 2))
    out = []
    for k in range(7):
        sc = vs if k < 3 else float(np.max(np.abs(want[k])))
        d = float(np.max(np.abs(np.asarray(got[k]) - np.asarray(want[k]))))
        out.append(0.0 if d == 0 else d / sc if sc > 0 else float("inf"))
    return out


def elem_rel_errors(got, want, floor=1e-3):
    """Per-property ELEMENT-relative error of two 7-field lists: max over cells of |a-b| / max(|b|, floor*scale),
    scale = max|b| of that property itself (so V and W are measured against their own magnitude, not against
    U's).  The floor keeps cells where a field passes through zero from dividing rounding noise by nothing."""
    out = []
    for k in range(7):
        a, b = np.asarray(got[k], np.float64), np.asarray(want[k], np.float64)
        d = np.abs(a - b)
        if not d.any():
            out.append(0.0)
            continue
        scale = float(np.max(np.abs(b)))
        if scale == 0:
            out.append(float("inf"))
            continue
        out.append(float(np.max(d / np.maximum(np.abs(b), floor * scale))))
    return out


def own_scale_errors(got, want, floor=1e-2):
    """Per-property error relative to THAT property's own largest magnitude: max|a-b| / max|b_k|.  A velocity
    component smaller than `floor` times the largest velocity component is measured against that floor instead:
    the components of one vector share their absolute rounding noise (every flux of V carries U), so a V that is
    1e-5 of U cannot be resolved to 1e-12 of itself by any arithmetic -- the reference's own is no exception (a
    one-ulp change of the viscosity moves such a V by 1e-9 of itself, profiles/r02/error_metrics_fast_sweep.txt)."""
    vs = max(float(np.max(np.abs(want[k]))) for k in (0, 1, 2))
    out = []
    for k in range(7):
        d = float(np.max(np.abs(np.asarray(got[k]) - np.asarray(want[k]))))
        sc = float(np.max(np.abs(want[k])))
        if k < 3:
            sc = max(sc, floor * vs)
        out.append(0.0 if d == 0 else d / sc if sc > 0 else float("inf"))
    return out
