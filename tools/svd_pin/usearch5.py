import numpy as np, torch, itertools
from run_search import run, ref
f32=np.float32
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
def r(x): return x.astype(f32)
print(run(["-DROTV=1","-DSROTV=1","-DLW=1","-DLU=1","-DRW=1","-DRU=1","-DRT=0"],"full"))
dbg=np.fromfile("dbg.bin",dtype=f32).reshape(-1,14)
UB=dbg[:,:9].reshape(-1,3,3).copy(); v1a,v1b,v2,t1,t2=[dbg[:,i] for i in (9,10,11,12,13)]
U=ref["full"][0].numpy(); n=UB.shape[0]
C=[[UB[:,i,j] for j in range(3)] for i in range(3)]
# H2 (pinned): w = c1 + r(c2*v2); c1 = fma(-t2,w,c1); c2 = fma(v2, r(-t2*w), c2)
A=[[C[i][j] for j in range(3)] for i in range(3)]
for j in range(3):
    w=r(C[1][j]+r(C[2][j]*v2))
    A[1][j]=fma(-t2,w,C[1][j]); A[2][j]=fma(v2,r(-t2*w),C[2][j])
res=[]
for wm in ['pair','seq','fma','pairfma','pairfma2']:
    out=[[None]*3 for _ in range(3)]
    for j in range(3):
        c=[A[i][j] for i in range(3)]
        if wm=='pair': w=r(c[0]+r(r(c[1]*v1a)+r(c[2]*v1b)))
        elif wm=='seq': w=r(r(c[0]+r(c[1]*v1a))+r(c[2]*v1b))
        elif wm=='fma': w=fma(c[2],v1b,fma(c[1],v1a,c[0]))
        elif wm=='pairfma': w=r(c[0]+fma(c[2],v1b,r(c[1]*v1a)))
        elif wm=='pairfma2': w=r(c[0]+fma(c[1],v1a,r(c[2]*v1b)))
        t=r(-t1*w)
        out[0][j]=fma(-t1,w,c[0]); out[1][j]=fma(v1a,t,c[1]); out[2][j]=fma(v1b,t,c[2])
    Uc=np.stack([np.stack(out[i],1) for i in range(3)],1); m=(Uc==U)
    res.append((float(m.reshape(n,-1).all(1).mean()),wm,np.round(m.mean(0),3).tolist()))
res.sort(key=lambda x:-x[0])
for x in res: print(x)
