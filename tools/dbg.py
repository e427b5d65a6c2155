import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T,B,N,L=400,64,40,30
dev="cuda:0"
g=torch.Generator().manual_seed(0)
tr=torch.rand(N,N,generator=g).to(dev); x=torch.randn(T,B,N,generator=g).to(dev); tg=torch.randint(0,N,(B,L),generator=g).to(dev)
il=torch.full((B,),T,dtype=torch.int64,device=dev); tl=torch.full((B,),L,dtype=torch.int64,device=dev)
be=torch_asg_amd.asg.native()
gf=torch.full((B,),1.0/B,device=dev); ga=-gf
for _ in range(3):
    full,ali,st=be.forward(x,tg,tr,il,tl,2)
    be.backward(st,gf,ga,x,tg,tr,il,tl)
torch.cuda.synchronize()
d=st[-512:-256].view(torch.int64).cpu().tolist()
print("stamps deltas:", [d[i]-d[i-1] for i in range(1,32) if d[i]])
print("total", max(d)-d[0])
