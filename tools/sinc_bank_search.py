import itertools
G0=[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]
G1=[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
GROUPS=[G0,G1,[l+32 for l in G0],[l+32 for l in G1]]
REG=4608//16
def conflicts(rowslot):
    worst=0
    for grp in GROUPS:
        seen={}
        for l in grp:
            r=l&15; kq=l>>4
            a=rowslot[r]+kq
            seen.setdefault(a%16,set()).add(a)
        worst=max(worst,max(len(v) for v in seen.values()))
    return worst
def st1(R):
    def f(p):
        j=p&3
        byte=20*p+((16-4*j)%16)
        return byte//16+R[j]+j*REG
    return f
def goodquads(f,nslots):
    good=[]
    for q in itertools.permutations(range(nslots),4):
        ok=True
        for b in range(3):
            rs=[f(12*q[r>>2]+4*b+(r&3)) for r in range(16)]
            if conflicts(rs)>1: ok=False;break
        if ok: good.append(q)
    return good
def part(good,rem,acc):
    if not rem: return acc
    first=min(rem)
    for q in good:
        if first in q and set(q)<=rem:
            r=part(good,rem-set(q),acc+[q])
            if r: return r
    return None
best=[]
for R in itertools.product(range(16),repeat=3):
    RR=(0,)+R
    g=goodquads(st1(RR),8)
    if g: best.append((len(g),RR))
best.sort(reverse=True)
print(best[:10], len(best))
for cnt,RR in best[:10]:
    g=goodquads(st1(RR),16)
    p=part(g,set(range(16)),[])
    print(RR,len(g),p)
    if p: break
