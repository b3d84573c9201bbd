"""Diagnostic: crude in-order issue model of each basic block of a kernel (`hipcc -S --cuda-device-only` listing): VALU 4.4
cycles (v_readlane 8.6, transcendental 16), LDS ops return in order LAT cycles after issue (b64 64, b128 80; one issued per 8 /
16 cycles), `s_waitcnt lgkmcnt(n)` stalls until at most n are outstanding.  Prints per block the modelled cycles, the part of
them spent stalled on LDS, and the number of waits that stalled: a ranking of where a lone wave sits out LDS round trips.
usage: isa_stall_model.py file.s <mangled kernel> <min loop depth> [min stall cycles]"""
import sys, re, collections

f, name = sys.argv[1:3]
mind = int(sys.argv[3]); minstall = float(sys.argv[4]) if len(sys.argv) > 4 else 200
lines = open(f).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(name + ':')][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
blocks = collections.OrderedDict(); depth = {}; bb = 'entry'; depth[bb] = 0
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)', l)
    if m:
        bb = m.group(1); d = re.search(r'Depth=(\d+)', m.group(2)); depth[bb] = int(d.group(1)) if d else 0
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    blocks.setdefault(bb, []).append(t)
rows = []
for b, ins in blocks.items():
    if depth[b] < mind: continue
    t = 0.0; lds_free = 0.0; q = []; stall = 0.0; nst = 0
    for i in ins:
        op = i.split()[0]
        if op.startswith('ds_'):
            wide = ('b128' in op) or ('read2' in op) or ('write2' in op)
            issue = max(t, lds_free); lds_free = issue + (16 if wide else 8)
            q.append(issue + (80 if wide else 64)); t += 4
        elif op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', i)
            if m and q:
                n = int(m.group(1))
                if len(q) > n:
                    done = q[len(q) - n - 1]
                    if done > t: stall += done - t; nst += 1; t = done
                    q = q[len(q) - n:] if n else []
        elif op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): t += 8.6
        elif op.startswith(('v_rcp', 'v_sqrt', 'v_rsq', 'v_sin', 'v_cos', 'v_exp', 'v_log')): t += 16
        elif op.startswith('v_'): t += 4.4
        elif op.startswith('s_'): t += 1
        else: t += 4
    rows.append((stall, b, depth[b], len(ins), t, nst))
for stall, b, d, n, t, nst in sorted(rows, reverse=True):
    if stall >= minstall: print(f"{b:12s} depth {d} instrs {n:4d} modelled {t:7.0f} cycles, stalled on LDS {stall:6.0f} ({100*stall/t:3.0f} %), {nst} stalling waits")
