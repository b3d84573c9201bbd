"""Occupancy experiment for the 4-wavefront srbd37 kernel: what would two workgroups per CU buy?
Run under rocprofv3 --kernel-trace --stats with a diagnostic library (SDDP_LIB): one backward sweep of B srbd37 instances
(N = 20; the phase-level entry point takes at most one instance per resident slot: 256, or 512 in the `half` build), three
times.  Builds compared:
  base : the shipped backward_kernel_mw (512 registers, LDS 115.7 KB -> one workgroup per CU)
  w2   : -DSDDP_EXP_W2_BACKWARD (256 registers), full LDS -> still one workgroup per CU: the register penalty alone
  half : -DSDDP_EXP_W2_BACKWARD -DSDDP_EXP_HALF_LDS -DSDDP_EXP_NO_PIVOT_EXIT: Q aliased onto Vxx / F~^T so that the tiles take
         79.3 KB and two workgroups share a CU.  The instruction stream is the shipped one, the NUMBERS ARE GARBAGE (and the
         pivot-failure exit is compiled out so that every knot still runs): a timing probe, nothing else.
usage: python3 profiles/r03/experiments/exp_mw_occupancy.py B waves_per_simd"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine

B, N, WPS = int(sys.argv[1]), 20, int(sys.argv[2])
b = workload.make_batch("srbd37", N, np.arange(B))
e = DdpEngine("srbd37", N, B, opts=dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, waves_per_simd=WPS))
print("slots", e.queue_info())
e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"])
for _ in range(3):
    kff, K, scal = e.backward(b["params"], mu=1e-3)
print("ok fraction", float(np.mean(scal[:, 4])))
