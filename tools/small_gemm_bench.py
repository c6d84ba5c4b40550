"""Isolated timings of the f32-MFMA GEMMs with few rows (the FFN head: one row per molecule): forward layer (K = 301, bias,
ReLU, dropout) and masked dX + dZ side output at M = 2048 / 4096 / 8192.  RR_LIB_PATH selects a library build."""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from reactranker_amd import functions as Fn
dev="cuda"; torch.manual_seed(0)
def t(fn, n=50, reps=5):
    for _ in range(10): fn()
    out=[]
    for _ in range(reps):
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1)/n*1e3)
    return statistics.median(out)
for M in (4096, 2048, 8192):
    H=300
    x=torch.randn(M,304,device=dev); x[:,301:]=0
    W=torch.randn(H,301,device=dev)/17; b=torch.randn(H,device=dev)
    L=Fn.LinW(W,b,big=False); wp=L.pk(301)
    out=torch.empty(M,H,device=dev)
    us=t(lambda: Fn.linear(M,H,wp,w_packed=True,a1=x,k1=301,bias=b,act=Fn.ACT_RELU,drop_p=0.1,seed=3,out=out))
    y=torch.relu(torch.randn(M,H,device=dev)); dz=torch.empty(M,H,device=dev)
    L2=Fn.LinW(torch.randn(H,H,device=dev)/17,b,big=False); wt=L2.pk_t(0,H)
    us2=t(lambda: Fn.linear(M,H,wt,w_packed=True,a1=x[:,:300] if False else y,k1=H,a_mask=y,mask_scale=1.1,out=out,dz_out=dz))
    print(f"M {M}: fwd K301 relu dropout {us:.1f} us   dX masked + dz {us2:.1f} us", flush=True)
