"""The C-ABI library loads, exports every symbol include/cfd_hip.h declares, and fails loudly
(no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from cfd_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include", "cfd_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(cfd_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_capi.EXPORTS) == syms


def test_default_params_are_the_shipped_configuration():
    lib = _capi.load()
    p = _capi.CfdParams()
    assert lib.cfd_default_params(ctypes.byref(p)) == 0
    assert (p.nx, p.ny, p.nz) == (301, 151, 1)
    assert p.dx == 1.6 / 301 and p.dy == 0.8 / 151 and p.dz == 1.0
    assert p.dt == 0.00001 and p.R == 289.0 and p.cv == 716.0 and p.k == 0.025 and p.dye_f == 0.001
    assert p.lambda_ == 0.010299990531505463
    assert lib.cfd_default_params(None) == _capi.CFD_E_ARG


def test_params_struct_layout_matches_header():
    # 3 x int32 (+4 pad), 10 doubles, 4 x int32
    assert ctypes.sizeof(_capi.CfdParams) == 16 + 80 + 16


def test_argument_errors_do_not_need_a_gpu():
    lib = _capi.load()
    assert lib.cfd_sweep(None, 1) == _capi.CFD_E_ARG
    assert b"null" in lib.cfd_last_error()
    assert lib.cfd_step(None, 1) == _capi.CFD_E_ARG
    assert lib.cfd_time_filter(None) == _capi.CFD_E_ARG


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cfd_amd import cases
    from cfd_amd.solver import AdjGraph
    params, fields, active = cases.synthetic_2d(8, 8)
    with pytest.raises(_capi.CfdError) as e:
        AdjGraph(params, fields, active)
    assert e.value.code == _capi.CFD_E_DEVICE


def test_compiled_host_fails_loudly_without_gpu():
    """The C++ host links against the C ABI alone; with no usable GPU it must exit non-zero with the
    library's error text, not fall back to anything."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "driver", "cfd_main")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "driver")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=100)
    assert out.returncode != 0
    assert "cfd_create" in out.stderr


def test_placement_of_blocks_beside_a_running_launch():
    """The host arithmetic that places a slab's late blocks on the XCDs with compute units left (a pure function of the
    library, no GPU): workgroup i of the running launch sits on XCD (i + oa) % 8, workgroup b of the late launch on
    (b + ob) % 8; the blocks that take work are numbered 0 .. units-1 in launch order, no XCD gets more than it has
    room for, and nothing is placed when the room does not suffice."""
    import ctypes
    import random
    from cfd_amd import _capi
    lib = _capi.load()
    fill = lib.cfd_internal_xcd_fill
    fill.restype = ctypes.c_int
    fill.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    rng = random.Random(5)
    out = (ctypes.c_int * 4096)()
    cases = [(34, 221, 32, 7, 0), (17, 238, 32, 0, 3), (34, 221, 32, 0, 0)]
    cases += [(rng.randint(1, 60), rng.randint(0, 256), 32, rng.randint(0, 7), rng.randint(0, 7)) for _ in range(300)]
    for units, busy, cap, oa, ob in cases:
        used = [sum(1 for i in range(busy) if (i + oa) % 8 == x) for x in range(8)]
        free = [max(0, cap - u) for u in used]
        grid = fill(units, busy, cap, oa, ob, out, 4096)
        if sum(free) < units:
            assert grid == 0
            continue
        assert 0 < grid <= 4096
        ranks = [out[b] for b in range(grid)]
        real = [r for r in ranks if r >= 0]
        assert real == list(range(len(real))) and len(real) == units and ranks[-1] == units - 1
        per_xcd = [0] * 8
        for b, r in enumerate(ranks):
            if r >= 0:
                per_xcd[(b + ob) % 8] += 1
        assert all(per_xcd[x] <= free[x] for x in range(8)), (units, busy, oa, ob, per_xcd, free)
