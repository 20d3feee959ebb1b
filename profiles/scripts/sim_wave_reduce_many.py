import numpy as np
def fold32(a,b):
    out=np.empty(64,dtype=object)
    for l in range(32): out[l]=a[l]+a[l+32]
    for l in range(32,64): out[l]=b[l-32]+b[l]
    return out
def fold16(x,y):
    out=np.empty(64,dtype=object)
    for l in range(64):
        row=l//16
        if row==0: out[l]=x[l]+x[l+16]
        elif row==1: out[l]=y[l-16]+y[l]
        elif row==2: out[l]=x[l]+x[l+16]
        else: out[l]=y[l-16]+y[l]
    return out
def xorstep(r,s):
    n=len(r)
    if n%2: r=r+[np.array([frozenset()]*64,dtype=object)]; n+=1
    out=[]
    for i in range(n//2):
        a,b=r[2*i],r[2*i+1]
        o=np.empty(64,dtype=object)
        for l in range(64):
            keep = b[l] if (l&s) else a[l]
            p=l^s
            send_from_partner = a[p] if (p&s)==0 and False else None
            # partner sends: partner lane p computes send = (p&s)? a[p] : b[p]
            send = a[p] if (p&s) else b[p]
            o[l]=keep+send
        out.append(o)
    return out
class S:
    # symbolic multiset: dict (value,lane)->count
    def __init__(s,d=None): s.d=d or {}
    def __add__(s,o):
        d=dict(s.d)
        for k,v in o.d.items(): d[k]=d.get(k,0)+v
        return S(d)
def run(N):
    NP=(N+3)//4*4
    r=[np.array([S({(i,l):1}) if i<N else S() for l in range(64)],dtype=object) for i in range(NP)]
    r=[fold32(r[2*i],r[2*i+1]) for i in range(NP//2)]
    r=[fold16(r[2*i],r[2*i+1]) for i in range(NP//4)]
    for s in (8,4,2,1):
        if len(r)==1:
            a=r[0]; o=np.empty(64,dtype=object)
            for l in range(64): o[l]=a[l]+a[l^s]
            r=[o]
        else:
            # generic
            n=len(r)
            if n%2: r=r+[np.array([S() for _ in range(64)],dtype=object)]
            out=[]
            for i in range(len(r)//2):
                a,b=r[2*i],r[2*i+1]; o=np.empty(64,dtype=object)
                for l in range(64):
                    p=l^s
                    keep=b[l] if (l&s) else a[l]
                    send=a[p] if (p&s) else b[p]
                    o[l]=keep+send
                out.append(o)
            r=out
    assert len(r)==1
    idx=[]
    for l in range(64):
        d=r[0][l].d
        vals=set(k[0] for k in d)
        if not vals: idx.append(-1); continue
        assert len(vals)==1,(l,vals)
        v=vals.pop()
        assert sorted(k[1] for k in d)==list(range(64)) and all(c==1 for c in d.values()),(l,v)
        idx.append(v)
    return idx
for N in (6,12,15,16,21,24,27,30,32):
    idx=run(N)
    print(N, idx)
