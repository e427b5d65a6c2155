import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, B, N, L = 400, 64, 40, 30
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)
tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
m = torch_asg_amd.ASGLoss(N, launch_mode=sys.argv[1]).to(dev)
for _ in range(6):
    m.transition.grad = None; x.grad = None
    m(x, tg, il, tl).backward()
torch.cuda.synchronize()
