import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, B, N, L = 400, 64, 40, 30
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev); tg = torch.randint(0, N, (B, L), generator=g).to(dev)
il = torch.full((B,), T, dtype=torch.int64, device=dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
be = torch_asg_amd.asg.native()
for _ in range(3): full, ali, st = be.forward(x, tg, tr, il, tl, 2)
torch.cuda.synchronize()
v = st[: st.numel() // 8 * 8].view(torch.int64).cpu()
idx = (v == 0x1234567890abcdef).nonzero().flatten()
for i in idx.tolist():
    d = v[i:i + 5].tolist()
    nb = d[4]
    h = v[i + 8:i + 11].tolist()
    print("helper: loop %d cycles, waiting for main %d, refill %d (%.0f/block)" % (h[0], h[1], h[2], h[2] / max(nb, 1)))
    print("loop cycles %d over %d blocks = %.0f/block (%.1f/step); poll %.0f/block; steps %.0f/block (%.1f/step)" % (d[1], nb, d[1] / nb, d[1] / nb / 16, d[2] / nb, d[3] / nb, d[3] / nb / 16))
