import numpy as np, torch, itertools
torch.manual_seed(0)
n=100000
F=(torch.eye(3)[None]+0.5*torch.randn(n,3,3)).contiguous()
A,tau=torch.geqrf(F)          # MKL sgeqrf
A=A.numpy(); tau=tau.numpy(); Fn=F.numpy()
f32=np.float32
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
a0,x0,x1=Fn[:,0,0],Fn[:,1,0],Fn[:,2,0]
def nrm2(v,x0,x1):
    if v=='ssq':
        a0_,a1_=np.abs(x0),np.abs(x1)
        # element 1: scale=a0, ssq=1 (if x0!=0)
        scale=a0_.copy(); ssq=np.ones_like(scale)
        z=(a0_==0); scale[z]=0
        big=(scale<a1_)&(a1_!=0)
        q=np.where(big, scale/np.where(a1_==0,1,a1_), a1_/np.where(scale==0,1,scale)).astype(f32)
        ssq=np.where(big, (f32(1)+ssq*(q*q)).astype(f32), np.where(a1_!=0,(ssq+q*q).astype(f32),ssq)).astype(f32)
        scale=np.where(big,a1_,scale)
        return (scale*np.sqrt(ssq)).astype(f32)
    if v=='plain': return np.sqrt((x0*x0+x1*x1).astype(f32)).astype(f32)
    if v=='fma01': return np.sqrt(fma(x1,x1,(x0*x0).astype(f32)))
    if v=='fma10': return np.sqrt(fma(x0,x0,(x1*x1).astype(f32)))
    if v=='dbl': return np.sqrt(x0.astype(np.float64)**2+x1.astype(np.float64)**2).astype(f32)
def lapy2(v,a,b):
    aa,bb=np.abs(a),np.abs(b); w=np.maximum(aa,bb); z=np.minimum(aa,bb)
    if v=='ref':
        q=(z/w).astype(f32); return (w*np.sqrt((f32(1)+q*q).astype(f32))).astype(f32)
    if v=='fma':
        q=(z/w).astype(f32); return (w*np.sqrt(fma(q,q,np.ones_like(q)))).astype(f32)
    if v=='plain': return np.sqrt((a*a+b*b).astype(f32))
    if v=='fmaab': return np.sqrt(fma(b,b,(a*a).astype(f32)))
    if v=='fmaba': return np.sqrt(fma(a,a,(b*b).astype(f32)))
    if v=='dbl': return np.sqrt(a.astype(np.float64)**2+b.astype(np.float64)**2).astype(f32)
R00=A[:,0,0]
best=[]
for nv in ['ssq','plain','fma01','fma10','dbl']:
    xn=nrm2(nv,x0,x1)
    for lv in ['ref','fma','plain','fmaab','fmaba','dbl']:
        beta=-np.copysign(lapy2(lv,a0,xn),a0)
        best.append((float((beta==R00).mean()),nv,lv))
# direct 3-norm variants
for name,val in [('n3plain',np.sqrt(((a0*a0+x0*x0).astype(f32)+x1*x1).astype(f32))),
                 ('n3fma',np.sqrt(fma(x1,x1,fma(x0,x0,(a0*a0).astype(f32))))),
                 ('n3dbl',np.sqrt(a0.astype(np.float64)**2+x0.astype(np.float64)**2+x1.astype(np.float64)**2).astype(f32))]:
    beta=-np.copysign(val,a0); best.append((float((beta==R00).mean()),name,''))
best.sort(reverse=True)
for b in best[:12]: print(b)
print("---- tau / scal")
xn=nrm2('ssq',x0,x1); beta=-np.copysign(lapy2('ref',a0,xn),a0)
tau1=((beta-a0)/beta).astype(f32)
print("tau ref", float((tau1==tau[:,0]).mean()))
for name,v in [('recip',(x0*(f32(1)/(a0-beta)).astype(f32)).astype(f32)),('div',(x0/(a0-beta)).astype(f32))]:
    print("v scal",name,float((v==A[:,1,0]).mean()))
v1=(x0/(a0-beta)).astype(f32); v2=(x1/(a0-beta)).astype(f32)
print("---- left slarf: R01 (row 0, col 1) and a11', a21' are not all visible; R01,R02 visible")
def gemv(v,c0,c1,c2):
    if v=='ref': return ((c0+(c1*v1).astype(f32)).astype(f32)+(c2*v2).astype(f32)).astype(f32)
    if v=='fma': return fma(c2,v2,fma(c1,v1,c0))
    if v=='fma2': return fma(c1,v1,fma(c2,v2,c0))
    if v=='pair': return (c0+((c1*v1).astype(f32)+(c2*v2).astype(f32)).astype(f32)).astype(f32)
    if v=='pairfma': return (c0+fma(c2,v2,(c1*v1).astype(f32))).astype(f32)
    if v=='dbl': return (c0.astype(np.float64)+c1.astype(np.float64)*v1+c2.astype(np.float64)*v2).astype(f32)
    if v=='rev': return (((c2*v2).astype(f32)+(c1*v1).astype(f32)).astype(f32)+c0).astype(f32)
def ger(v,c,vr,w,tau):
    if v=='ref': t=(-tau*w).astype(f32); return (c+(vr*t).astype(f32)).astype(f32)
    if v=='fma': t=(-tau*w).astype(f32); return fma(vr,t,c)
    if v=='tv': t=(-tau*vr).astype(f32); return (c+(t*w).astype(f32)).astype(f32)
    if v=='tvfma': t=(-tau*vr).astype(f32); return fma(t,w,c)
    if v=='sub': t=(tau*w).astype(f32); return (c-(vr*t).astype(f32)).astype(f32)
    if v=='subfma': t=(tau*w).astype(f32); return fma(-vr,t,c)
    if v=='dbl': return (c.astype(np.float64)-tau.astype(np.float64)*w.astype(np.float64)*vr.astype(np.float64)).astype(f32)
one=np.ones_like(a0)
res=[]
for gv in ['ref','fma','fma2','pair','pairfma','dbl','rev']:
    for rv in ['ref','fma','tv','tvfma','sub','subfma','dbl']:
        ok=np.ones(n,bool)
        for j in (1,2):
            w=gemv(gv,Fn[:,0,j],Fn[:,1,j],Fn[:,2,j])
            r0=ger(rv,Fn[:,0,j],one,w,tau1)
            ok&=(r0==A[:,0,j])
        res.append((float(ok.mean()),gv,rv))
res.sort(reverse=True)
for r in res[:10]: print(r)
print("---- ger variant via step 2")
def nrm1(x): return np.abs(x)
for rv in ['ref','fma','tv','tvfma','sub','subfma','dbl']:
    cols={}
    for j in (1,2):
        w=gemv('ref',Fn[:,0,j],Fn[:,1,j],Fn[:,2,j])
        cols[j]=[ger(rv,Fn[:,r,j],[one,v1,v2][r],w,tau1) for r in range(3)]
    a11,a21=cols[1][1],cols[1][2]
    xn2=np.abs(a21)
    beta2=-np.copysign(lapy2('ref',a11,xn2),a11)
    beta2=np.where(a21==0,a11,beta2)
    tau2=np.where(a21==0,0,((beta2-a11)/beta2)).astype(f32)
    vv=np.where(a21==0,0,(a21/(a11-beta2))).astype(f32)
    okb=(beta2==A[:,1,1]); okt=(tau2==tau[:,1]); okv=(vv==A[:,2,1])
    # apply to column 2 rows 1,2
    c1,c2=cols[2][1],cols[2][2]
    res2=[]
    for gv2 in ['ref','fma']:
        w2=(c1+(c2*vv).astype(f32)).astype(f32) if gv2=='ref' else fma(c2,vv,c1)
        for rv2 in ['ref','fma','tv','tvfma']:
            r12=ger(rv2,c1,one,w2,tau2); r22=ger(rv2,c2,vv,w2,tau2)
            res2.append((float(((r12==A[:,1,2])&(r22==A[:,2,2])).mean()),gv2,rv2))
    res2.sort(reverse=True)
    print(rv,"beta2",float(okb.mean()),"tau2",float(okt.mean()),"v",float(okv.mean()),res2[:3])
