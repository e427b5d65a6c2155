#!/usr/bin/env python3
"""Developer probe (GPU): full-lattice forward alone (asg_full_forward: batched kernel + its clean-up launch) at a few batch sizes,
for the library named by ASG_HIP_LIB (tools/devbuild_batched.sh variants).  Prints us per call (graph replay of 5 calls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch_asg_amd
T, N = 400, 40
dev = "cuda:0"
be = torch_asg_amd.asg.native()
out = []
for B in [int(a) for a in sys.argv[1:]] or [512, 4096]:
    g = torch.Generator().manual_seed(0)
    tr = torch.rand(N, N, generator=g).to(dev); x = torch.randn(T, B, N, generator=g).to(dev)
    il = torch.full((B,), T, dtype=torch.int64, device=dev)
    L = 30
    tg = torch.randint(0, N, (B, L), generator=g).to(dev); tl = torch.full((B,), L, dtype=torch.int64, device=dev)
    if os.environ.get("ASG_ABL_WHAT", "full") == "aligned":
        def fn(): be.aligned_forward(x, tg, tr, il, tl)
    else:
        def fn(): be.full_forward(x, tr, il)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(5): fn()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): gr.replay()
    torch.cuda.synchronize(); out.append("B=%d %.1f us" % (B, (time.perf_counter() - t0) / 10 / 5 * 1e6))
print("%-12s %-8s %s" % (os.path.basename(os.environ.get("ASG_HIP_LIB", "shipped")), os.environ.get("ASG_ABL_WHAT", "full"), "   ".join(out)))
