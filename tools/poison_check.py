import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch_asg_amd, util
dev = "cuda:0"
def run(T, B, N, L, dtype, seed):
    rng = np.random.default_rng(seed)
    tr, x, tg, _, _ = util.synth(T, B, N, L, N)
    il = rng.integers(max(1, T // 2), T + 1, B); tl = np.minimum(rng.integers(1, L + 1, B), il)
    m = torch_asg_amd.ASGLoss(N, reduction="none").to(dev).to(dtype)
    with torch.no_grad(): m.transition.copy_(tr.to(dtype))
    xd = x.to(dev, dtype).requires_grad_(True)
    loss = m(xd, tg.to(dev), torch.from_numpy(il).to(dev), torch.from_numpy(tl).to(dev)); loss.sum().backward(); torch.cuda.synchronize()
    g = m.transition.grad.cpu().numpy()
    bad = ~np.isfinite(g)
    print(T, B, N, L, dtype, "loss", loss.detach().cpu().numpy()[:3], "nonfinite grad_transition:", int(bad.sum()), "rows", np.unique(np.nonzero(bad)[0])[:10], "cols", np.unique(np.nonzero(bad)[1])[:10])
# poison the allocator's free memory
junk = torch.full((600 * 1024 * 1024 // 4,), float("nan"), device=dev); del junk
run(7, 3, 2100, 3, torch.float32, 2100)
junk = torch.full((600 * 1024 * 1024 // 4,), float("inf"), device=dev); del junk
run(7, 3, 2100, 3, torch.float32, 2100)
run(6, 40, 1100, 2, torch.float32, 1100)
