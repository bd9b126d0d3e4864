import sys; sys.path.insert(0,"."); sys.argv=["x"]
import torch
import tools.bench_gemm8p as b
ok = b.check(8192, 2048, 8192, zmask=True, bias=False, scale=0.5) and b.check(4099, 3072, 768, act=4, resid=True, zmask=True) and b.check(8192, 8192, 2048, act=3, resid=True) and b.check(40960, 2048, 2048, act=1)
print("OK" if ok else "FAIL")
from mmgl_amd import ops
M,N,K=40960,8192,2048
x=torch.randn(M,K,device="cuda").bfloat16(); W=(torch.randn(N,K,device="cuda")*K**-0.5).bfloat16(); z=torch.randn(M,N,device="cuda").bfloat16(); y=torch.empty(M,N,device="cuda",dtype=torch.bfloat16)
for rep in range(2):
  for name,kw in (("plain",{}),("zmask",dict(zmask=z)),("resid",dict(residual=z))):
    for _ in range(3): ops.gemm_nt(x,W,out=y,**kw)
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.gemm_nt(x,W,out=y,**kw)
    e.record(); torch.cuda.synchronize()
    t=s.elapsed_time(e)/10*1e-3
    print(name, round(t*1e6,1),"us", round(2*M*N*K/t/1e12,1),"TF")
