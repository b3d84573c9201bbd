#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) accesses sit?  Reads a `hipcc -O3 --offload-arch=gfx950 -gline-tables-only -S --cuda-device-only`
listing, finds the loops of one kernel (backward branches) and counts scratch loads / stores and their source lines per loop body.

    python tools/isa_loop_scratch.py listing.s <mangled kernel name> <min loop length> <max loop length>

Round 4 (profiles/r04/experiments/README.md): solve_kernel_w2<srbd13> has 4 scratch loads in its sweep knot
loop and none in the rollout knot loop; solve_kernel_mw<srbd61> about 150 and 61."""
import re, collections, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
fl = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    fl[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
labels = {}; ins = []; cur = None
for l in s[i:j].splitlines():
    t = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m: cur = (fl.get(m.group(1), '?'), int(m.group(2))); continue
    m = re.match(r'^(\.LBB[\w_]+):', t)
    if m: labels[m.group(1)] = len(ins); continue
    if not t or t[0] in '.;': continue
    ins.append((t, cur))
loops = []
for k, (t, loc) in enumerate(ins):
    m = re.match(r's_cbranch\w*\s+(\.LBB[\w_]+)|s_branch\s+(\.LBB[\w_]+)', t)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= k: loops.append((labels[lab], k))
loops = sorted(set(loops), key=lambda x: x[1] - x[0], reverse=True)
print("instrs", len(ins), "scratch", sum(1 for t, _ in ins if t.startswith('scratch_')))
lo, hi = int(sys.argv[3]), int(sys.argv[4])
for a, b in loops:
    n = b - a + 1
    if not (lo <= n <= hi): continue
    body = ins[a:b + 1]
    sc = [(t, loc) for t, loc in body if t.startswith('scratch_')]
    locs = collections.Counter((loc[0][:14], loc[1] // 50 * 50) for _, loc in body if loc)
    sloc = collections.Counter((loc[0][:14], loc[1] // 10 * 10, t.split('_')[1]) for t, loc in sc if loc)
    print(f"loop {a}-{b} len {n} scratch ld {sum(1 for t,_ in sc if 'load' in t)} st {sum(1 for t,_ in sc if 'store' in t)}; lines {locs.most_common(2)}; scratch at {sloc.most_common(5)}")
