"""A/B of PMOE_RES_INBN (BatchNorm + ReLU of the input applied on load by conv3x3_respipe_kernel<false, 3>) against the plain launch
and the pmoe_bn_apply pass it replaces, at the U-Net level-1 shape of BASELINE config 4.  python tools/ab_inbn.py"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from pmoe_amd import ops, hip
DEV="cuda"
BF=torch.bfloat16
N,H,W,C=64,256,256,64
z=torch.randn(N,H,W,C,device=DEV).to(BF)
w=(torch.randn(1,64,9,64,device=DEV)*0.05).to(BF)
out=torch.empty_like(z)
coef=torch.rand(4,1,C,device=DEV)+0.5
kw=dict(cin=C,cout=C,coutp=C,ipe=N,ks=3,stride=1,pad=1)
rows=ops.conv2d_stat_rows(N,H,W,H,W,C,C,C,N,3,1,1,BF)
st=torch.zeros(rows,2,C,device=DEV)
def t(f,n=20):
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(True),torch.cuda.Event(True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for rep in range(2):
    print("plain      %.3f ms" % t(lambda: ops.conv2d(z,w,out,stats=st,**kw)))
    print("inbn       %.3f ms" % t(lambda: ops.conv2d(z,w,out,stats=st,res_mode=hip.RES_INBN,bn_coef=coef,**kw)))
a_=torch.empty_like(z)
print("bn_apply   %.3f ms" % t(lambda: ops.bn_apply(z,None,a_,coef[2],coef[3],coef[0],N*H*W,1,C,True)))
