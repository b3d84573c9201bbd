"""Diagnostic: per-phase shader-cycle shares of the fused solve kernel (build with -DSDDP_STAMPS, SDDP_LIB=...)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbd_horizon_amd import workload, _lib
from srbd_horizon_amd.engine import DdpEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MODEL = sys.argv[2] if len(sys.argv) > 2 else "srbd13"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30
SO = int(sys.argv[4]) if len(sys.argv) > 4 else 1            # sddp_options.second_order
WPS = int(sys.argv[5]) if len(sys.argv) > 5 else 1           # sddp_options.waves_per_simd
batch = workload.make_batch(MODEL, N, np.arange(B))
eng = DdpEngine(MODEL, N, B, opts=dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, second_order=SO, waves_per_simd=WPS), consts=batch["consts"])
eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
eng.enable_timing(True)
x, u = eng.solve(batch["params"])
sc = np.zeros((B, 24))
fn = eng.lib.sddp_debug_read_scal; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p]
assert fn(eng.h, sc.ctypes.data_as(C.c_void_p)) == 0
it = eng.stats["iters"]; ro = eng.stats["rollouts"]
names = ["derivs", "bw.stage", "bw.expand+vp", "bw.W (W-free: GC|WC + gathers)", "bw.Q=H+FtW", "bw.solve", "bw.Vupd+gains", "-", "rollout", "other"]
names += ["ro.feedback", "ro.close-knot", "ro.step", "bw.Q.blocks(mw)", "bw.Q.qv+barrier(mw)", "s15", "gj.load", "gj.owner", "gj.syncwait", "gj.update(+last)", "gj.publish", "s21","s22","s23"]
tot = sc[:, :24].sum(axis=1)
print("kernel ms", eng.last_kernel_ms(), "mean iters", it.mean(), "mean rollouts", ro.mean())
print("cycles/iter (mean over instances): %.0f" % (tot / np.maximum(it, 1)).mean())
for i, n in enumerate(names):
    print("  %-14s %6.1f%%  %10.0f cyc/iter" % (n, 100 * sc[:, i].sum() / tot.sum(), (sc[:, i] / np.maximum(it, 1)).mean()))
