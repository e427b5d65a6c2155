#!/usr/bin/env python3
"""Developer probe (GPU): does a hipMemsetAsync recorded into a hipGraph still write its value when the graph is REPLAYED?
Round 6 found the 256-byte memset of the loss ticket replaying as garbage (its own node parameters) on ROCm 7.2 -- a replayed stand-alone
step kept its first loss.  Per arrangement (memset first in the graph / behind a kernel / several memsets) and size: is the region zero after
each of three replays that follow a fill with ones?"""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = "cuda:0"
print("torch", torch.__version__, torch.cuda.get_device_name(0))


def run(arrangement, sizes):
    bufs = [torch.ones(n // 4 + 64, dtype=torch.int32, device=dev) for n in sizes]
    other = torch.zeros(64, device=dev)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if arrangement != "memset first":
                other.add_(1)
            for b, n in zip(bufs, sizes):
                assert hip.hipMemsetAsync(b.data_ptr(), 0, n, st) == 0
                if arrangement == "kernel between memsets":
                    other.add_(1)
            other.add_(1)
    out = []
    for rep in range(3):
        for b in bufs:
            b.fill_(1)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ok = [int(b[: n // 4].abs().sum()) == 0 for b, n in zip(bufs, sizes)]
        out.append("".join("Z" if o else "x" for o in ok))
    print("%-24s sizes %-28s replays (Z = zero, x = garbage, one letter per memset): %s" % (arrangement, sizes, " | ".join(out)))


for arr in ("memset first", "kernel before", "kernel between memsets"):
    run(arr, [256])
    run(arr, [1 << 20])
    run(arr, [256, 4096, 1 << 20])
