"""Initial conditions: the reference's shipped case in closed form, and the synthetic grids
the benchmarks and parity tests run on.

Every generator returns (params, fields, active):
  params  dict for solver.AdjGraph / make_params
  fields  list of 7 float64 arrays in propIndex order (U, V, W, Density, Temperature, Mew, Dye),
          node order x slowest, z fastest
  active  uint8 array, 1 = the node's `active` flag
"""
import numpy as np

from .solver import make_params

# reference values (src/SolutionDomain.hs:136-148, :240-244, :264-277)
REF_NX, REF_NY, REF_NZ = 301, 151, 1
REF_DX, REF_DY, REF_DZ = 1.6 / 301, 0.8 / 151, 1.0 / 1
U0, RHO0, T0, MEW0 = 10.0, 0.3, 223.0, 0.1


def _haskell_round(x):
    """Haskell `round`: to nearest, ties to even -- numpy's rint."""
    return int(np.rint(x))


def reference_case(**kw):
    """The shipped configuration, initialGrid_adj (src/SolutionDomain.hs:262-353), without the
    Map/Dijkstra machinery: 301 x 151 x 1; U = 10, V = W = 0, Density = 0.3, Temperature = 223,
    Mew = 0.1; Dye = 1 at (x = 0, y = 75); inactive nodes = the inflow column x = 0
    (inflowPositions, :379-380) and the filled square whose corners squareBoundsPts (:214-224)
    rounds to; U = V = W = 0 inside the square (initialGrid, :292-306)."""
    nx, ny, nz = REF_NX, REF_NY, REF_NZ
    gx, gy = 1.6, 0.8
    # corners through coordToPos (:160-166): round(coord / cellLength)
    x0 = _haskell_round((gx / 4) / REF_DX)
    x1 = _haskell_round((gx / 4 + 0.1) / REF_DX)
    y0 = _haskell_round((gy / 2 - 0.05) / REF_DY)
    y1 = _haskell_round((gy / 2 + 0.05) / REF_DY)
    n = nx * ny * nz
    U = np.full((nx, ny, nz), U0)
    V = np.zeros((nx, ny, nz))
    W = np.zeros((nx, ny, nz))
    rho = np.full((nx, ny, nz), RHO0)
    T = np.full((nx, ny, nz), T0)
    mew = np.full((nx, ny, nz), MEW0)
    dye = np.zeros((nx, ny, nz))
    dye[0, (ny - 1) // 2, :] = 1.0
    active = np.ones((nx, ny, nz), np.uint8)
    active[0, :, :] = 0
    active[x0:x1 + 1, y0:y1 + 1, :] = 0
    for f in (U, V, W):
        f[x0:x1 + 1, y0:y1 + 1, :] = 0.0
    params = make_params(nx, ny, nz, REF_DX, REF_DY, REF_DZ, **kw)
    return params, [a.reshape(n) for a in (U, V, W, rho, T, mew, dye)], active.reshape(n)


def _obstacle_box(nx, ny):
    """The reference square scaled by index fraction."""
    x0, x1 = int(75 / 301 * nx), int(94 / 301 * nx)
    y0, y1 = int(66 / 151 * ny), int(85 / 151 * ny)
    return x0, max(x1, x0), y0, max(y1, y0)


def synthetic_2d(nx, ny, seed=1, **kw):
    """Reference case scaled to nx x ny x 1 with the reference cell size, inflow column, scaled
    obstacle, and a perturbation so that the limiter takes every branch:
    Density*(1+1e-3 u), Temperature*(1+1e-3 u), U + 0.1 u, V = 0.1 u, u ~ U(-1,1)."""
    rng = np.random.default_rng(seed)
    n = nx * ny
    u = lambda: rng.uniform(-1.0, 1.0, n)
    rho = RHO0 * (1 + 1e-3 * u())
    T = T0 * (1 + 1e-3 * u())
    U = U0 + 0.1 * u()
    V = 0.1 * u()
    W = np.zeros(n)
    mew = np.full(n, MEW0)
    dye = np.zeros(n)
    dye.reshape(nx, ny)[0, (ny - 1) // 2] = 1.0
    active = np.ones((nx, ny), np.uint8)
    active[0, :] = 0
    x0, x1, y0, y1 = _obstacle_box(nx, ny)
    active[x0:x1 + 1, y0:y1 + 1] = 0
    for f in (U, V):
        f.reshape(nx, ny)[x0:x1 + 1, y0:y1 + 1] = 0.0
    params = make_params(nx, ny, 1, REF_DX, REF_DY, REF_DZ, **kw)
    return params, [U, V, W, rho, T, mew, dye], active.reshape(n)


def cavity_2d(nx, ny, seed=2, **kw):
    """Lid-driven-cavity style case: a one-cell inactive ring; lid row y = ny-1 moves with U = 10,
    the other walls and the whole interior have U = V = 0 exactly (so the NaN -> 1 rule of the
    unknown coefficient fires everywhere); Density = 0.3, Temperature = 223*(1 + 1e-6 u)."""
    rng = np.random.default_rng(seed)
    n = nx * ny
    U = np.zeros((nx, ny))
    U[:, ny - 1] = U0
    V = np.zeros((nx, ny))
    W = np.zeros((nx, ny))
    rho = np.full((nx, ny), RHO0)
    T = T0 * (1 + 1e-6 * rng.uniform(-1.0, 1.0, (nx, ny)))
    mew = np.full((nx, ny), MEW0)
    dye = np.zeros((nx, ny))
    dye[nx // 2, ny // 2] = 1.0
    active = np.ones((nx, ny), np.uint8)
    active[0, :] = active[nx - 1, :] = 0
    active[:, 0] = active[:, ny - 1] = 0
    params = make_params(nx, ny, 1, REF_DX, REF_DY, REF_DZ, **kw)
    return params, [a.reshape(n) for a in (U, V, W, rho, T, mew, dye)], active.reshape(n)


def cavity_2d_slab(nx_global, ny, x0, x1, seed=2, **kw):
    """Planes x0 .. x1-1 of cavity_2d(nx_global, ny, seed), built without the global arrays (the
    random stream is advanced to the slab's first value), for one rank of a slab decomposition."""
    from numpy.random import PCG64, Generator
    bg = PCG64(seed)
    bg.advance(x0 * ny)                     # one 64-bit draw per uniform double
    rng = Generator(bg)
    sx = x1 - x0
    n = sx * ny
    U = np.zeros((sx, ny))
    U[:, ny - 1] = U0
    V = np.zeros((sx, ny))
    W = np.zeros((sx, ny))
    rho = np.full((sx, ny), RHO0)
    T = T0 * (1 + 1e-6 * rng.uniform(-1.0, 1.0, (sx, ny)))
    mew = np.full((sx, ny), MEW0)
    dye = np.zeros((sx, ny))
    if x0 <= nx_global // 2 < x1:
        dye[nx_global // 2 - x0, ny // 2] = 1.0
    active = np.ones((sx, ny), np.uint8)
    if x0 == 0:
        active[0, :] = 0
    if x1 == nx_global:
        active[sx - 1, :] = 0
    active[:, 0] = active[:, ny - 1] = 0
    params = make_params(sx, ny, 1, REF_DX, REF_DY, REF_DZ, **kw)
    return params, [a.reshape(n) for a in (U, V, W, rho, T, mew, dye)], active.reshape(n)


def synthetic_3d(nx, ny, nz, seed=3, **kw):
    """synthetic_2d's recipe in three dimensions: dz = dx, W = 0.1 u, obstacle spans the middle
    third in z."""
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    u = lambda: rng.uniform(-1.0, 1.0, n)
    rho = RHO0 * (1 + 1e-3 * u())
    T = T0 * (1 + 1e-3 * u())
    U = U0 + 0.1 * u()
    V = 0.1 * u()
    W = 0.1 * u()
    mew = np.full(n, MEW0)
    dye = np.zeros(n)
    dye.reshape(nx, ny, nz)[0, (ny - 1) // 2, :] = 1.0
    active = np.ones((nx, ny, nz), np.uint8)
    active[0, :, :] = 0
    x0, x1, y0, y1 = _obstacle_box(nx, ny)
    z0, z1 = nz // 3, max(nz // 3, 2 * nz // 3)
    active[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = 0
    for f in (U, V, W):
        f.reshape(nx, ny, nz)[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = 0.0
    params = make_params(nx, ny, nz, REF_DX, REF_DY, REF_DX, **kw)
    return params, [U, V, W, rho, T, mew, dye], active.reshape(n)


def smooth_2d(nx, ny, dt=1e-7, **kw):
    """Smooth, well-resolved fields (a few sine modes, amplitudes of a few per cent) and a small
    time step: a configuration in which the explicit scheme does not amplify rounding differences,
    used to measure drift between arithmetic variants over many steps.  No inactive cells."""
    x = (np.arange(nx) + 0.5)[:, None] / nx
    y = (np.arange(ny) + 0.5)[None, :] / ny
    two_pi = 2.0 * np.pi
    U = U0 * (1 + 0.02 * np.sin(two_pi * x) * np.cos(two_pi * y))
    V = 0.2 * np.cos(two_pi * x) * np.sin(two_pi * 2 * y) + 0.05
    rho = RHO0 * (1 + 0.01 * np.cos(two_pi * x) * np.cos(two_pi * y))
    T = T0 * (1 + 0.01 * np.sin(two_pi * 2 * x) * np.sin(two_pi * y))
    dye = 0.5 + 0.4 * np.sin(two_pi * x) * np.sin(two_pi * y)
    n = nx * ny
    W = np.zeros(n)
    mew = np.full(n, MEW0)
    active = np.ones(n, np.uint8)
    params = make_params(nx, ny, 1, REF_DX, REF_DY, REF_DZ, dt=dt, **kw)
    return params, [U.reshape(n), V.reshape(n), W, rho.reshape(n), T.reshape(n), mew, dye.reshape(n)], active
