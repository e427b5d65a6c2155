#!/usr/bin/env python3
"""Build libasg_hip.so for gfx950 with hipcc (no torch headers, no cmake).

    python torch_asg_amd/csrc/build.py [--force] [--verbose]

Each .hip translation unit is compiled to an object in parallel, then linked into
torch_asg_amd/csrc/libasg_hip.so (git-ignored; travels to the GPU box with the snapshot).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["asg_small_f32.hip", "asg_small_f64.hip", "asg_bwd_f32.hip", "asg_bwd_f64.hip", "asg_fused.hip", "asg_generic.hip", "asg_generic_step.hip", "asg_generic_aligned.hip", "asg_generic_grad.hip",
           "asg_viterbi.hip", "asg_api.hip"]
HEADERS = ["asg_common.h", "asg_kernels.h", "asg_generic_common.h", "asg_chains.h", "asg_outer.h", "asg_assemble.h", "asg_small_impl.inc", "asg_bwd_impl.inc",
           os.path.join("..", "..", "include", "asg_hip.h")]
OUT = os.path.join(HERE, "libasg_hip.so")
ARCH = os.environ.get("ASG_HIP_ARCH", "gfx950")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (-Wno-inline-asm: the LDS-DMA transfers of the contraction set M0 inside their asm statement and say so in the clobber list, which
# clang reports as "reserved register" once per statement)
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
          "-ffp-contract=off"]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: developer experiments -- build a variant library next to the real one."""
    global OUT
    if defines or out:
        import tempfile
        odir = tempfile.mkdtemp(prefix="asgvar_")
        objs = []
        procs = []
        for s in SOURCES:
            obj = os.path.join(odir, s.replace(".hip", ".o"))
            objs.append(obj)
            procs.append(subprocess.Popen([HIPCC] + CFLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(HERE, s), "-o", obj],
                                          stderr=subprocess.DEVNULL))
        for p in procs:
            assert p.wait() == 0
        subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
        return out
    hdr_t = max(_mtime(os.path.join(HERE, h)) for h in HEADERS)
    hdr_t = max(hdr_t, _mtime(__file__))
    procs, objs = [], []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _mtime(obj) < max(_mtime(src), hdr_t):
            cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    if procs or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


# Developer variants of the library that the GPU test-suite loads through ASG_HIP_LIB (tests/test_hip_variants.py):
# ONE translation unit differs, every other object is the shipped one.
VARIANTS = {
    "spread": ("asg_fused.hip", ["ASG_X_SPREAD_XCD"]),
    "delay": ("asg_fused.hip", ["ASG_X_TEST_DELAY=60000", "ASG_X_CAPSHIFT=7"]),
    # workgroup 1 of every cluster of fwd_cluster_kernel never publishes a frame: its peers' bounded waits must run out
    # (2^16 polls in this build) and poison the scores with NaN -- no hang, no wrong number
    "stall": ("asg_generic_step.hip", ["ASG_X_CL_TEST_STALL", "ASG_X_CL_SPINMAX=65536"]),
}
VAR_DIR = os.path.join(HERE, "var_libs")


def build_variants(force=False, verbose=False):
    """var_libs/libasg_hip_<name>.so for every entry of VARIANTS (git-ignored; they travel to the GPU box)."""
    build(force=False, verbose=verbose)
    os.makedirs(VAR_DIR, exist_ok=True)
    hdr_t = max([_mtime(__file__)] + [_mtime(os.path.join(HERE, h)) for h in HEADERS])
    procs = []
    for name, (srcname, defs) in VARIANTS.items():
        src = os.path.join(HERE, srcname)
        obj = os.path.join(VAR_DIR, "%s_%s.o" % (srcname.replace(".hip", ""), name))
        if force or _mtime(obj) < max(_mtime(src), hdr_t):
            cmd = [HIPCC] + CFLAGS + ["-D" + d for d in defs] + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((name, subprocess.Popen(cmd, stderr=None if verbose else subprocess.DEVNULL)))
    for name, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on variant %s" % name)
    outs = []
    for name, (srcname, defs) in VARIANTS.items():
        out = os.path.join(VAR_DIR, "libasg_hip_%s.so" % name)
        objs = [os.path.join(HERE, s_.replace(".hip", ".o")) for s_ in SOURCES if s_ != srcname]
        objs.append(os.path.join(VAR_DIR, "%s_%s.o" % (srcname.replace(".hip", ""), name)))
        if force or _mtime(out) < max(_mtime(o) for o in objs):
            subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
        outs.append(out)
    return outs


BINDING_SRC = os.path.join(HERE, "binding.cpp")
BINDING_OUT = os.path.join(HERE, "..", "_binding.so")


def build_binding(force=False, verbose=False):
    """torch_asg_amd/_binding.so: the C++ host fast path of ASGLossFunction (binding.cpp) -- g++ against the torch
    headers of the running interpreter; reaches libasg_hip.so only through addresses handed over at init()."""
    hdr = os.path.join(HERE, "..", "..", "include", "asg_hip.h")
    if not force and _mtime(BINDING_OUT) > max(_mtime(BINDING_SRC), _mtime(hdr), _mtime(__file__)):
        return BINDING_OUT
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_binding", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi())]
    cmd += ["-I" + d for d in ce.include_paths()] + ["-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]]
    cmd += [BINDING_SRC, "-o", BINDING_OUT, "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
            "-ltorch_python", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return BINDING_OUT


if __name__ == "__main__":
    if "--binding" in sys.argv:
        print(build_binding(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
        sys.exit(0)
    if "--variants" in sys.argv:
        print("\n".join(build_variants(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)))
        sys.exit(0)
    if "--define" in sys.argv:
        i = sys.argv.index("--define")
        defs = sys.argv[i + 1].split(",")
        print(build(defines=[d for d in defs if d], out=sys.argv[sys.argv.index("--out") + 1]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
    print(build_binding(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
