import torch, time
import torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev="cuda:0"
B,H,N,D=256,12,197,64
qkv=torch.randn(B,N,3,H,D,device=dev,dtype=torch.float16)
q,k,v=qkv.permute(2,0,3,1,4).unbind(0)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for name,be in (("flash",SDPBackend.FLASH_ATTENTION),("efficient",SDPBackend.EFFICIENT_ATTENTION),("math",SDPBackend.MATH)):
    try:
        with sdpa_kernel(be):
            ms=t(lambda: F.scaled_dot_product_attention(q,k,v,scale=D**-0.5))
        print(name, "%.3f ms"%ms)
    except Exception as ex:
        print(name,"failed",str(ex)[:100])
qc,kc,vc=[x.contiguous() for x in (q,k,v)]
with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
    print("flash contiguous qkv %.3f ms"%t(lambda: F.scaled_dot_product_attention(qc,kc,vc,scale=D**-0.5)))
# padded N=208/256
for NP in (208,256):
    qp=torch.randn(B,H,NP,D,device=dev,dtype=torch.float16)
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        print("flash N=%d %.3f ms"%(NP,t(lambda: F.scaled_dot_product_attention(qp,qp,qp,scale=D**-0.5))))
# manual bmm path
def manual():
    a=(q@k.transpose(-2,-1))*(D**-0.5); a=a.softmax(-1); return a@v
print("manual bmm %.3f ms"%t(manual))
