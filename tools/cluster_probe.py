#!/usr/bin/env python3
"""Developer probe (GPU, ASG_X_CL_PROBE build: tools/devbuild_generic.sh clprobe -DASG_X_CL_PROBE, ASG_HIP_LIB=.../variants/libclprobe.so):
forwards of the resident-slice kernel per shape T,B,N given on the command line; the kernel prints its phases (cycles per frame);
also the forward's time by events (meaningful with the product library too).
ASG_DTYPE=f64: double precision."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch_asg_amd, util
dev = "cuda:0"
for a in sys.argv[1:] or ["400,64,512"]:
    T, B, N = (int(v) for v in a.split(","))
    tr, x, tg, il, tl = util.synth(T, B, N, 30, 0, True)
    if os.environ.get("ASG_DTYPE") == "f64":
        tr, x = tr.double(), x.double()
    be = torch_asg_amd.asg.native()
    print("T=%d B=%d N=%d %s" % (T, B, N, x.dtype), flush=True)
    args = (x.to(dev), tg.to(dev), tr.to(dev), il.to(dev), tl.to(dev), 0)
    for _ in range(2):
        be.forward(*args)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        be.forward(*args)
    e1.record(); torch.cuda.synchronize()
    print("  forward (both lattices, one stream): %.3f ms" % (e0.elapsed_time(e1) / 5), flush=True)
