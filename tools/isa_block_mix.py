"""Diagnostic: instruction-class mix per basic block (with loop depth) of a kernel in a `hipcc -S --cuda-device-only` listing.
usage: isa_block_mix.py file.s <mangled kernel> <min instructions>"""
import sys,re,collections
f=sys.argv[1]; name=sys.argv[2]; minn=int(sys.argv[3])
lines=open(f).read().split('\n')
start=[i for i,l in enumerate(lines) if l.startswith(name+':')][0]
end=[i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end')][0]
body=lines[start:end]
bb='entry'; blocks=collections.OrderedDict(); info={}
for l in body:
    m=re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)',l)
    if m: bb=m.group(1); info[bb]=m.group(2).strip()
    if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        blocks.setdefault(bb,[]).append(l.strip())
def cls(op):
    if re.match(r'v_(fma|mul|add|fmac)_f64',op): return 'fp64'
    if 'f64' in op: return 'f64x'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch_'): return 'scr'
    if op.startswith(('global_','flat_','buffer_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('v_readlane','v_readfirstlane','v_writelane')): return 'lane'
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('v_mov'): return 'vmov'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_cmp'): return 'cmp'
    return 'valu'
tot=collections.Counter()
for b,ins in blocks.items():
    c=collections.Counter(cls(i.split()[0]) for i in ins)
    d=re.search(r'Depth=(\d+)',info.get(b,''))
    h=re.search(r'Header=(\w+)',info.get(b,''))
    if len(ins)>=minn:
        print(b,len(ins),'depth',d.group(1) if d else '-', h.group(1) if h else '-', dict(c))
