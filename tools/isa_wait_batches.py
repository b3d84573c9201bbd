"""Diagnostic: for each basic block, the LDS-wait profile: number of s_waitcnt on lgkmcnt, and for each the number of LDS reads
issued since the previous wait (batch size).  Many waits with batch size 1-2 = serialised LDS round trips.
usage: isa_wait_batches.py file.s <mangled kernel> <min loop depth>"""
import sys,re,collections
f,name=sys.argv[1:3]; mind=int(sys.argv[3])
lines=open(f).read().split('\n')
start=[i for i,l in enumerate(lines) if l.startswith(name+':')][0]
end=[i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end')][0]
bb='entry'; depth={'entry':0}; batches=collections.defaultdict(list); size=collections.Counter(); pend=0
for l in lines[start:end]:
    m=re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)',l)
    if m:
        bb=m.group(1); d=re.search(r'Depth=(\d+)',m.group(2)); depth[bb]=int(d.group(1)) if d else 0; pend=0
        continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    size[bb]+=1
    op=t.split()[0]
    if op.startswith('ds_read'): pend+=1
    m=re.search(r'lgkmcnt\((\d+)\)',t)
    if op=='s_waitcnt' and m:
        left=int(m.group(1))
        if pend>0 and left==0:
            batches[bb].append(pend); pend=0
for b,v in batches.items():
    small=sum(1 for x in v if x<=2)
    if depth.get(b,0)>=mind and small>=3:
        print(b,'depth',depth[b],'instrs',size[b],'full waits',len(v),'with <=2 reads in flight',small,'batches',v[:40])
