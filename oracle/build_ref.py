#!/usr/bin/env python3
"""Build the REAL reference CPU path into oracle/_ref/ (test infrastructure only).

This compiles zh217/torch-asg's own C++ sources *where they lie* under
/root/reference (torch_asg/native/{utils,force_aligned_lattice,
fully_connected_lattice,extension}.cpp -- the CppExtension source list of
/root/reference/setup.py:36-41, same flags `-fopenmp -Ofast`) with plain g++
against the libtorch headers of the installed torch.  No reference source is
copied into this repository: each translation unit is streamed to g++ on stdin.

One token has to change for a modern ATen: fully_connected_lattice.cpp:60 calls
`.sum({0, 1})`, which is ambiguous between the IntArrayRef and DimnameList
overloads in torch >= 1.3.  The stream edit below rewrites exactly that call to
`.sum(at::IntArrayRef({0, 1}))` on the fly (nothing is written back anywhere).

Output: oracle/_ref/torch_asg_native.so (a pybind11 module, git-ignored, travels
to the GPU box with the snapshot).  It exports the reference's four CPU entry
points (extension.cpp:16-19).  Nothing outside tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may load it.

If /root/reference is absent (GPU box) this script is a no-op.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("ASG_REFERENCE_ROOT", "/root/reference")
NATIVE = os.path.join(REF_ROOT, "torch_asg", "native")
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "torch_asg_native.so")
SOURCES = ["utils.cpp", "force_aligned_lattice.cpp", "fully_connected_lattice.cpp", "extension.cpp"]


def _flags():
    import torch
    from torch.utils import cpp_extension as ce
    inc = []
    for p in ce.include_paths() + [sysconfig.get_paths()["include"]]:
        inc += ["-isystem", p]
    try:
        import pybind11
        inc += ["-isystem", pybind11.get_include()]
    except Exception:
        pass
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cflags = ["-std=c++17", "-fPIC", "-fopenmp", "-Ofast", "-w",
              "-DTORCH_EXTENSION_NAME=torch_asg_native",
              "-DTORCH_API_INCLUDE_EXTENSION_H",
              "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi,
              "-I", NATIVE] + inc
    libdir = ce.library_paths()[0]
    ldflags = ["-shared", "-fopenmp", "-L", libdir, "-Wl,-rpath," + libdir,
               "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    return cflags, ldflags


def _stream_edit(name, text):
    if name == "fully_connected_lattice.cpp":
        old = ".sum({0, 1})"
        assert text.count(old) == 1, "reference changed: expected exactly one '.sum({0, 1})'"
        text = text.replace(old, ".sum(at::IntArrayRef({0, 1}))")
    return text


def up_to_date():
    if not os.path.exists(OUT_SO):
        return False
    if not os.path.isdir(NATIVE):
        return True
    newest = max(os.path.getmtime(os.path.join(NATIVE, s)) for s in SOURCES)
    return os.path.getmtime(OUT_SO) >= max(newest, os.path.getmtime(__file__))


def build(force=False, verbose=True):
    if not os.path.isdir(NATIVE):
        if verbose:
            print("[oracle/_ref] %s absent: keeping prebuilt %s" % (REF_ROOT, OUT_SO))
        return os.path.exists(OUT_SO)
    if up_to_date() and not force:
        return True
    os.makedirs(OUT_DIR, exist_ok=True)
    cflags, ldflags = _flags()
    objs = []
    procs = []
    for s in SOURCES:
        with open(os.path.join(NATIVE, s)) as f:
            text = _stream_edit(s, f.read())
        obj = os.path.join(OUT_DIR, s.replace(".cpp", ".o"))
        objs.append(obj)
        cmd = ["g++", "-x", "c++", "-c", "-", "-o", obj] + cflags
        p = subprocess.Popen(cmd, stdin=subprocess.PIPE)
        p.stdin.write(text.encode())
        p.stdin.close()
        procs.append((s, p))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("reference build failed on %s" % s)
    subprocess.check_call(["g++"] + objs + ["-o", OUT_SO] + ldflags)
    for o in objs:
        os.remove(o)
    if verbose:
        print("[oracle/_ref] built", OUT_SO)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    sys.exit(0 if ok else 1)
