import numpy as np, torch, itertools, subprocess
f32=np.float32
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
def r(x): return x.astype(f32)
torch.manual_seed(5)
n=100000
F=(torch.eye(3)[None]+0.5*torch.randn(n,3,3)).contiguous(); F[:,1,0]=0; F[:,2,0]=0
F.numpy().tofile("h2only.bin"); U=torch.svd(F)[0].numpy()
subprocess.check_call(["g++","-O2","-std=c++17","-ffp-contract=off","-o","drv","-DROTV=1","-DSROTV=1","-DLW=1","-DLU=1","-DRW=1","-DRU=1","-DRT=0","drv.cpp"],stderr=subprocess.DEVNULL)
subprocess.check_call(["./drv","h2only.bin","out.bin"])
dbg=np.fromfile("dbg.bin",dtype=f32).reshape(-1,14)
UB=dbg[:,:9].reshape(-1,3,3).copy(); v1a,v1b,v2,t1,t2=[dbg[:,i] for i in (9,10,11,12,13)]
print("t1==0 fraction", float((t1==0).mean()))
one=np.ones(n,f32)
C=[[UB[:,i,j] for j in range(3)] for i in range(3)]
res=[]
for wm in ['plain','fma','fma_b']:
  for um in ['fma','plain','tv','tvfma','tw2','tw2fma']:
    out=[[C[i][j] for j in range(3)] for i in range(3)]
    for j in range(3):
        if wm=='plain': w=r(C[1][j]+r(C[2][j]*v2))
        elif wm=='fma': w=fma(C[2][j],v2,C[1][j])
        elif wm=='fma_b': w=fma(C[1][j],one,r(C[2][j]*v2))
        t=r(-t2*w)
        if um=='fma': o1=fma(one,t,C[1][j]); o2=fma(v2,t,C[2][j])
        elif um=='plain': o1=r(C[1][j]+t); o2=r(C[2][j]+r(v2*t))
        elif um=='tv': o1=r(C[1][j]+r(r(-t2)*w)); o2=r(C[2][j]+r(r(-t2*v2)*w))
        elif um=='tvfma': o1=fma(r(-t2),w,C[1][j]); o2=fma(r(-t2*v2),w,C[2][j])
        elif um=='tw2': tt=r(t2*w); o1=r(C[1][j]-tt); o2=r(C[2][j]-r(v2*tt))
        elif um=='tw2fma': tt=r(t2*w); o1=r(C[1][j]-tt); o2=fma(-v2,tt,C[2][j])
        out[1][j]=o1; out[2][j]=o2
    Uc=np.stack([np.stack(out[i],1) for i in range(3)],1); m=(Uc==U)
    res.append((float(m.reshape(n,-1).all(1).mean()),wm,um,np.round(m.mean(0),3).tolist()))
res.sort(key=lambda x:-x[0])
for x in res[:6]: print(x)
