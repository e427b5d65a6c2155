"""Developer probe (GPU): per-lattice gradient / score errors of the HIP path against the fp64 oracle at cfg sizes."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch_asg_amd
from oracle import asg_oracle as orc
dev = "cuda:0"
for (T, B, N, L, var) in ((400, 64, 40, 30, False), (400, 64, 40, 30, True), (150, 16, 30, 20, True)):
    g = torch.Generator().manual_seed(1)
    tr = torch.rand(N, N, generator=g); x = torch.randn(T, B, N, generator=g); tg = torch.randint(0, N, (B, L), generator=g)
    il = torch.randint(T // 2, T + 1, (B,), generator=g) if var else torch.full((B,), T)
    tl = torch.randint(L // 2, L + 1, (B,), generator=g) if var else torch.full((B,), L)
    xd64, trd64 = x.double().numpy(), tr.double().numpy()
    for nm, F, fw, bw in (("FCC", torch_asg_amd.FCC, lambda: orc.full_forward(xd64, trd64, il.numpy()), None), ("FAC", torch_asg_amd.FAC, None, None)):
        t_ = tr.to(dev).clone().requires_grad_(True); xx = x.to(dev).requires_grad_(True)
        s = F.apply(t_, xx, tg.to(dev), il.to(dev), tl.to(dev)); s.sum().backward()
        if nm == "FCC":
            sc, A1, B1 = orc.full_forward(xd64, trd64, il.numpy()); gtr, gin = orc.full_backward(np.ones(B), A1, B1, xd64, trd64)
        else:
            sc, A1, B1 = orc.aligned_forward(xd64, tg.numpy(), trd64, il.numpy(), tl.numpy()); gtr, gin = orc.aligned_backward(np.ones(B), A1, B1, tg.numpy(), trd64, il.numpy(), tl.numpy(), N)
        print(T, B, "var" if var else "fix", nm, "gi err %.2e  gt err %.2e  score err %.2e" % (np.abs(xx.grad.cpu().numpy() - gin).max(), np.abs(t_.grad.cpu().numpy() - gtr).max() / max(1, np.abs(gtr).max()), np.abs(s.detach().cpu().numpy() - sc).max() / np.abs(sc).max()))
