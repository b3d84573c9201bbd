"""Diagnostic: per basic block of a kernel in a `hipcc -S --cuda-device-only` listing, count the `s_waitcnt lgkmcnt(0)` that
directly follow a ds_read (an LDS round trip the wave sits out).  usage: isa_exposed_waits.py file.s <mangled kernel> <min loop depth>"""
import sys,re,collections
f,name=sys.argv[1:3]
lines=open(f).read().split('\n')
start=[i for i,l in enumerate(lines) if l.startswith(name+':')][0]
end=[i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end')][0]
bb='entry'; depth={}; cur=None
exposed=collections.Counter(); total=collections.Counter(); size=collections.Counter()
last_lds=-99; n=0
for l in lines[start:end]:
    m=re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)',l)
    if m:
        bb=m.group(1); d=re.search(r'Depth=(\d+)',m.group(2)); depth[bb]=int(d.group(1)) if d else 0
        continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    n+=1; size[bb]+=1
    op=t.split()[0]
    if op.startswith('ds_read'): last_lds=n
    if op=='s_waitcnt' and 'lgkmcnt(0)' in t:
        total[bb]+=1
        if n-last_lds<=2: exposed[bb]+=1
for b in size:
    if depth.get(b,0)>=int(sys.argv[3]) and exposed[b]>0:
        print(b,'depth',depth[b],'instrs',size[b],'lgkm0 waits',total[b],'right after a ds_read',exposed[b])
