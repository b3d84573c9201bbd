#!/usr/bin/env python3
"""Where does a kernel wait for memory?  Lists, in program order, every VMEM operation (global / scratch / buffer loads and stores),
every `s_waitcnt` that names vmcnt and every `s_barrier` of one kernel in a `hipcc -O3 --offload-arch=gfx950 -gline-tables-only -S
--cuda-device-only` listing, with the source line each belongs to -- optionally only inside an instruction-index window (the indices
tools/isa_loop_scratch.py prints for a loop).

    python tools/isa_vm_waits.py listing.s <mangled kernel name> [first last]

gfx9 counts loads and stores in ONE vmcnt; loads return in order, stores complete out of order with respect to loads, so with both
kinds in flight the compiler can only emit vmcnt(0).  Round 4 ("memory waits"): a `s_waitcnt vmcnt(0)` right
behind a knot's prefetch loads -- in front of a store, a spill reload, a zero fill of a masked load's register -- makes the prefetch
synchronous.  Blocks laid out of line (masked loads) appear where the layout puts them, not where they execute: read the branches."""
import re
import sys

f, name = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1 << 30)
s = open(f).read()
i = s.index(name + ":")
j = s.index(".Lfunc_end", i)
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
ins, cur = [], None
for line in s[i:j].splitlines():
    t = line.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m:
        cur = (files.get(m.group(1), '?'), int(m.group(2)))
        continue
    if not t or t[0] in '.;':
        continue
    ins.append((t, cur))
for n, (t, loc) in enumerate(ins):
    if not lo <= n <= hi:
        continue
    op = t.split()[0]
    if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_', 's_barrier')) or (op == 's_waitcnt' and 'vmcnt' in t):
        rest = t.split(None, 1)[1][:48] if ' ' in t else ''
        print(f"{n:6d}  {op:24s} {rest:50s} {loc[0] if loc else '':22s} {loc[1] if loc else ''}")
