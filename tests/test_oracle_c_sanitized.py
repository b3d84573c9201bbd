"""The plain-C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the CPU build only --
GPU ASan is not available on this pool): every model's knot evaluation and a short solve, in a child process with the sanitizer
runtime preloaded.  Catches what a plain run would not: stack arrays sized by the wrong model constant (oracle/c/srbd_cs.inc is
instantiated for nc = 4 and nc = 8), out-of-range contact indices, signed overflow in index arithmetic."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, {root!r})
from oracle import cport, ddp as oddp, models as omodels
cport._lib = C.CDLL({lib!r})
from srbd_horizon_amd import workload
opts = oddp.DdpOptions(max_iters=4, alpha_converge_threshold=1e-12, beta=1e-3)
for name, N in (("srbd13", 30), ("srbd37", 8), ("srbd61", 6), ("lip30", 20)):
    b = workload.make_batch(name, N, [1, 2])
    m = omodels.make_model(name)
    xs, us, st = cport.solve_batch(m.cst, opts, b["x0"], b["params"], b["xs"], b["us"], threads=2, model=name)
    assert np.all(np.isfinite(xs)) and np.all(np.isfinite(us)) and st[:, 1].min() >= 1, (name, st)
    for so in (0, 2):
        o2 = oddp.DdpOptions(max_iters=2, alpha_converge_threshold=1e-12, beta=1e-3, second_order=so)
        cport.solve_batch(m.cst, o2, b["x0"], b["params"], b["xs"], b["us"], threads=1, model=name)
    for k, term in ((0, False), (3, False), (N, True)):
        cport.eval_knot(m.cst, b["x0"][0], b["us"][0, 0], b["params"][0, min(k, N)], k, term, model=name)
bar = omodels.RobotConsts(friction_barrier_weight=2.0, friction_barrier_sharpness=4.0, bound_barrier_weight=1.0,
                          lower=np.full(19, -5.0), upper=np.full(19, 5.0))
b = workload.make_batch("srbd13", 30, [3])
cport.solve_batch(bar, opts, b["x0"], b["params"], b["xs"], b["us"], threads=1, model="srbd13")
print("sanitized oracle ok")
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_c_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    lib = str(tmp_path / "liboracle_san.so")
    subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-shared", "-o", lib, os.path.join(ROOT, "oracle", "c", "sddp_oracle.c"), "-lm"], check=True, capture_output=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], check=True, capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    env = dict(os.environ, LD_PRELOAD=asan + (":" + ubsan if os.path.isabs(ubsan) and os.path.exists(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, lib=lib)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
