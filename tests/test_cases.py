"""Initial-condition generators (host logic only)."""
import numpy as np

from cfd_amd import cases


def test_reference_case_geometry():
    params, f, active = cases.reference_case()
    nx, ny = params["nx"], params["ny"]
    a = active.reshape(nx, ny)
    assert (nx, ny, params["nz"]) == (301, 151, 1)
    assert not a[0].any()                                  # inflow column
    assert not a[75:95, 66:86].any() and int((a == 0).sum()) == 151 + 400
    U = f[0].reshape(nx, ny)
    assert np.all(U[75:95, 66:86] == 0) and np.all(U[0] == 10.0)
    dye = f[6].reshape(nx, ny)
    assert dye[0, 75] == 1.0 and dye.sum() == 1.0
    assert np.all(f[1] == 0) and np.all(f[2] == 0) and np.all(f[3] == 0.3) and np.all(f[4] == 223.0)


def test_synthetic_cases_are_seeded_and_shaped():
    p1, f1, a1 = cases.synthetic_2d(40, 30, seed=1)
    p2, f2, a2 = cases.synthetic_2d(40, 30, seed=1)
    assert all(np.array_equal(x, y) for x, y in zip(f1, f2)) and np.array_equal(a1, a2)
    assert np.all(f1[2] == 0) and f1[0].size == 1200
    pc, fc, ac = cases.cavity_2d(20, 20)
    assert np.all(fc[1] == 0) and int(ac.sum()) == 18 * 18
    assert np.all(fc[0].reshape(20, 20)[:, 19] == 10.0) and np.all(fc[0].reshape(20, 20)[:, :19] == 0)
    p3, f3, a3 = cases.synthetic_3d(10, 8, 6)
    assert f3[2].any() and p3["dz"] == p3["dx"] and f3[0].size == 480


def test_cavity_slabs_concatenate_to_the_global_case():
    from cfd_amd import cases
    nx, ny = 37, 11
    _, gf, ga = cases.cavity_2d(nx, ny)
    for world in (1, 2, 5):
        bounds = [nx * r // world for r in range(world + 1)]
        parts = [cases.cavity_2d_slab(nx, ny, bounds[r], bounds[r + 1]) for r in range(world)]
        for k in range(7):
            assert np.array_equal(np.concatenate([p[1][k] for p in parts]), gf[k]), (world, k)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), ga)
