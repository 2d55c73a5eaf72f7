import numpy as np, torch, subprocess, itertools, sys
torch.manual_seed(1)
n=100000
d=torch.randn(n,3)+1.0; e=torch.randn(n,2)*0.5
Fb=torch.zeros(n,3,3); Fb[:,0,0]=d[:,0];Fb[:,1,1]=d[:,1];Fb[:,2,2]=d[:,2];Fb[:,0,1]=e[:,0];Fb[:,1,2]=e[:,1]
Ff=(torch.eye(3)[None]+0.5*torch.randn(n,3,3)).contiguous()
ref={}
for name,F in (("bidiag",Fb),("full",Ff)):
    F.numpy().tofile(name+".bin"); ref[name]=torch.svd(F)
def run(flags, name):
    subprocess.check_call(["g++","-O2","-std=c++17","-ffp-contract=off","-o","drv"]+flags+["drv.cpp"],stderr=subprocess.DEVNULL)
    subprocess.check_call(["./drv",name+".bin","out.bin"])
    o=torch.from_numpy(np.fromfile("out.bin",dtype=np.float32).reshape(-1,21))
    u,s,v=o[:,:9].reshape(-1,3,3),o[:,9:12],o[:,12:].reshape(-1,3,3)
    U,S,V=ref[name]
    return float((S==s).all(1).float().mean()),float((U==u).flatten(1).all(1).float().mean()),float((V==v).flatten(1).all(1).float().mean())
if __name__=="__main__":
    for rv in range(5):
        for sv in range(5):
            print("ROTV",rv,"SROTV",sv,run(["-DROTV=%d"%rv,"-DSROTV=%d"%sv],"bidiag"),flush=True)
