import sys, math, numpy as np, torch
sys.path.insert(0,'/root/repo/bayesian-inference_amd'); sys.path.insert(0,'/root/repo')
from bayes_vi_amd import _native as nat
from oracle import closed_form as cf
dev=torch.device('cuda:0'); ws=nat.Workspace(dev)
for sigma in (0.1,0.01,0.001):
    rng=np.random.default_rng(17); n,d,c=200000,64,40
    X=rng.normal(size=(n,d)).astype(np.float32); w=rng.normal(size=d)
    y=(0.5+X.astype(np.float64)@w+sigma*rng.normal(size=n)).astype(np.float32)
    A=np.concatenate([np.ones((n,1)),X.astype(np.float64)],1); ols=np.linalg.lstsq(A,y.astype(np.float64),rcond=None)[0]
    width=sigma/math.sqrt(n)
    th=np.concatenate([ols[None,:1]+width*rng.normal(size=(c,1)),ols[None,1:]+width*rng.normal(size=(c,d)),sigma*rng.uniform(0.9,1.1,(c,1))],-1).astype(np.float32)
    ref_lp,ref_g=cf.linreg(X,y,th)
    Xd,yd,thd=[torch.tensor(a,device=dev) for a in (X,y,th)]
    pk=nat.linreg_pack(Xd,yd,fmt=nat.PACK_F16)
    lp,g=nat.linreg_packed_logp_grad(pk,n,d,thd,ws,fmt=nat.PACK_F16)
    lp32,g32=nat.linreg_logp_grad(Xd,yd,thd,ws,precision=nat.PREC_F32)
    e=lambda a:(np.abs(a.cpu().numpy()-ref_lp)/np.abs(ref_lp))
    print(sigma,'f16 logp rel max %.2e mean signed %.2e | f32 max %.2e mean signed %.2e'%(e(lp).max(),((lp.cpu().numpy()-ref_lp)/np.abs(ref_lp)).mean(),e(lp32).max(),((lp32.cpu().numpy()-ref_lp)/np.abs(ref_lp)).mean()))
