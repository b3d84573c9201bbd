"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/sddp.h declares (no compute calls without a GPU); the product path fails loudly without a device."""
import ctypes as C
import os
import re

import pytest

from srbd_horizon_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def test_header_symbols_are_all_bound_and_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "sddp.h")).read()
    declared = set(re.findall(r"\b(sddp_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layouts_and_dims(lib):
    assert lib.sddp_abi_version() == 9
    assert _lib.model_dims("srbd13") == (13, 6, 19)
    assert _lib.model_dims("srbd37") == (37, 24, 19)
    assert _lib.model_dims("lip30") == (30, 15, 11)
    assert _lib.model_dims("srbd61") == (61, 48, 27)                      # contact_model = 4, prb.py:39-41
    assert lib.sddp_model_dims(4, None, None, None) != 0
    o = _lib.default_options()
    # Python-side defaults of the reference adapter (ddp.py:17-29)
    assert (o.max_iters, o.alpha_0, o.alpha_converge_threshold, o.line_search_decrease_factor, o.beta) == (100, 1.0, 1e-1, 0.5, 1e-4)
    c = _lib.default_consts()
    assert c.force_scaling == 1000.0 and c.dt == 0.05 and c.inertia_mode == 0 and abs(c.com[2] - 0.88) < 1e-15
    assert C.sizeof(_lib.SddpStats) == 64                                 # v9: + rho
    # model-aware defaults (v9): srbd61's contact points 0..3 are the first four sole corners, not the line feet
    assert list(_lib.default_consts("srbd61").feet)[:6] == [0.08, 0.13, 0.0, -0.08, 0.13, 0.0]
    assert list(_lib.default_consts("srbd37").feet) == list(c.feet) == list(_lib.default_consts("srbd13").feet)


def test_set_consts_resets_the_tail_of_the_bounds():
    """a reused constants struct keeps no stale bound beyond the values given"""
    import numpy as np
    c = _lib.default_consts(lower=[-1.0] * 19, upper=[1.0] * 19)
    assert c.lower[18] == -1.0 and c.upper[18] == 1.0
    _lib.set_consts(c, lower=[-2.0] * 13, upper=[2.0] * 13)
    assert c.lower[12] == -2.0 and c.lower[13] == -np.inf and c.upper[18] == np.inf


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    rc = lib.sddp_create(C.byref(h), 0, 30, 1, None, None)
    assert rc != 0 and not h.value
    assert b"no HIP device" in lib.sddp_last_error(None)
    from srbd_horizon_amd.engine import DdpEngine
    with pytest.raises(RuntimeError):
        DdpEngine("srbd13", 30, 1)


def build_c_host(tmp_path):
    """examples/c_abi_solve.c: a plain-C host on include/sddp.h (gcc, no Python, no torch in the process)"""
    import subprocess
    exe = str(tmp_path / "c_abi_solve")
    libdir = os.path.join(ROOT, "srbd_horizon_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_solve.c"),
                    "-o", exe, "-L" + libdir, "-lsddp_hip", "-Wl,-rpath," + libdir, "-lm"], check=True, capture_output=True)
    return exe


def test_c_host_program_links_against_the_abi_and_fails_loudly_without_a_device(lib, tmp_path):
    import subprocess
    import torch
    exe = build_c_host(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: tests/test_gpu_shim.py runs the program")
    r = subprocess.run([exe, "4"], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr
