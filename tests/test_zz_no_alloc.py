"""cfd_total_density_noalloc (the entry point that allocates nothing: block partials in a fixed device
array of the library) against cfd_total_density, through the C ABI: two calls in a row, the second on a
smaller grid; bit-identical (same launch geometry)."""
import ctypes

import numpy as np
import pytest

from helpers import small_case

pytestmark = pytest.mark.gpu


def test_total_density_noalloc_matches_total_density():
    import torch
    from cfd_amd import _capi
    from cfd_amd.solver import AdjGraph
    lib = _capi.load()
    out = torch.full((2,), -1.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for i, (shape, flags) in enumerate([((300, 280, 1), 4), ((23, 17, 1), 0)]):
        params, fields, active = small_case(shape[0], shape[1], shape[2], 50 + i)
        g = AdjGraph(dict(params, flags=flags), fields, active)
        g.step(2)
        want = g.total_density()
        _capi.check(lib.cfd_total_density_noalloc(g._h, ctypes.c_void_p(out.data_ptr() + 8 * i)))
        g.sync()
        got = float(out[i].item())
        assert got == want and np.isfinite(got), (i, got, want)
        # again on the same handle: the array keeps the previous call's partials, which must not matter
        _capi.check(lib.cfd_total_density_noalloc(g._h, ctypes.c_void_p(out.data_ptr() + 8 * i)))
        g.sync()
        assert float(out[i].item()) == want
        assert lib.cfd_total_density_noalloc(g._h, None) == _capi.CFD_E_ARG
        g.close()
    assert lib.cfd_total_density_noalloc(None, ctypes.c_void_p(out.data_ptr())) == _capi.CFD_E_ARG
