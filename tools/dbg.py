import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T,B,N,L=400,64,40,30
dev="cuda:0"
g=torch.Generator().manual_seed(0)
tr=torch.rand(N,N,generator=g).to(dev); x=torch.randn(T,B,N,generator=g).to(dev); tg=torch.randint(0,N,(B,L),generator=g).to(dev)
il=torch.full((B,),T,dtype=torch.int64,device=dev); tl=torch.full((B,),L,dtype=torch.int64,device=dev)
be=torch_asg_amd.asg.native()
for _ in range(3):
    full,ali,st=be.forward(x,tg,tr,il,tl,2)
torch.cuda.synchronize()
off=2*B*T*N*4
d=st[off:off+24*5*8].view(torch.int64).cpu().reshape(24,5)
print("per block (cycles of s_memtime @100MHz?): loads-issue, steps, wait, copy ; block-to-block")
for i in range(1,24):
    a,b,c,dd,e=d[i].tolist()
    print(i, b-a, c-b, dd-c, e-dd, "| period", a-d[i-1][0].item())
