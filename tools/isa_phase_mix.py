#!/usr/bin/env python3
"""Instruction mix per PHASE of the one-wavefront solve kernel, from a listing with phase markers:

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Isrbd_horizon_amd/csrc -DSDDP_MARKS -DSDDP_INST_MODEL=Srbd13 \
          -DSDDP_INST_FN=ops_srbd13 '-DSDDP_INST_NAME="srbd13"' -S --cuda-device-only srbd_horizon_amd/csrc/sddp_inst.hip -o srbd13_marks.s
    python tools/isa_phase_mix.py srbd13_marks.s solve_kernel_w2

-DSDDP_MARKS turns the SDDP_TICK(i) phase boundaries of csrc/sddp_kernels.hpp into assembler comments.  Every instruction of the
kernel is attributed to the last marker in STATIC order; blocks are split by loop depth, so that the body of a knot loop (executed
once per knot: its inner loops are fully unrolled or run one trip) is counted apart from the code around it.  The counts are
wave-instructions per knot and phase -- what the SIMD has to issue -- beside the FMAs the algorithm needs at wave level."""
import collections
import re
import sys

PHASE = {"9>0": "derivatives (lane per knot, once per iteration)", "0>1": "sweep: set-up + terminal node | knot loop: staging",
         "1>2": "sweep knot: expand F~^T, v' = Vx + Vxx d", "2>3": "sweep knot: W = (V~ F~)^T", "3>4": "sweep knot: Q = D + F~^T W (+ torque term)",
         "4>5": "sweep knot: 6 x 6 Gauss-Jordan, gains, Vx", "5>6": "sweep knot: Vxx = Qxx + Qux^T K", "6>9": "sweep exit, line-search set-up",
         "9>8": "rollout (lane per step length): knot loop = feedback + model step + cost", "8>*": "accept / bookkeeping"}
# wave-level FMAs the algorithm needs per knot (srbd13; one lane's chain, the lanes run in parallel)
USEFUL = {"2>3": 84, "3>4": 72, "4>5": 6 * 6 + 6 + 6, "5>6": 12, "1>2": 14, "9>8": 78 + 150}


def cls(op):
    if re.match(r"v_(fma|mul|add|fmac)_f64", op): return "fp64"
    if "f64" in op: return "fp64x"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "lane"
    return "valu32"


def main():
    f, kname = sys.argv[1], sys.argv[2]
    lines = open(f).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kname + r"I\w*:", l)][0]
    end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
    depth, last, prev = 0, "*", "*"
    tab = collections.defaultdict(collections.Counter)          # (region, depth) -> class counts
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", l)
        if m:
            d = re.search(r"Depth=(\d+)", m.group(2))
            depth = int(d.group(1)) if d else 0
        m = re.search(r"; SDDP_MARK (\d+)", l)
        if m:
            prev, last = last, m.group(1)
            continue
        if l.startswith("\t") and not l.startswith(("\t.", "\t;")):
            tab[(last, depth)][cls(l.strip().split()[0])] += 1
    # a region is named by the marker that opens it and the one that closes it (static order)
    order = []
    for (r, d) in tab:
        if r not in order:
            order.append(r)
    closes = {r: (order[i + 1] if i + 1 < len(order) else "*") for i, r in enumerate(order)}
    cols = ["fp64", "fp64x", "valu32", "lane", "lds", "salu", "vmem", "scratch", "wait"]
    print(f"{kname}: wave-instructions by phase (static order) and loop depth")
    print(f"{'phase':78s} {'depth':>5s} " + " ".join(f"{c:>7s}" for c in cols) + f" {'total':>7s} {'useful FMA':>10s}")
    for (r, d), c in sorted(tab.items(), key=lambda kv: (order.index(kv[0][0]), kv[0][1])):
        name = f"{r}>{closes[r]}"
        tot = sum(c.values())
        if tot < 8:
            continue
        print(f"{PHASE.get(name, name):78s} {d:5d} " + " ".join(f"{c[k]:7d}" for k in cols) + f" {tot:7d} {USEFUL.get(name, ''):>10}")


if __name__ == "__main__":
    main()
