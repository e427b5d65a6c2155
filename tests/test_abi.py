"""CPU tests of the C-ABI boundary: libasg_hip.so loads without a GPU, exports every symbol that
include/asg_hip.h declares, sizes/validation behave -- no kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "asg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asg_[a-z_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = _declared_symbols()
    for n in ("asg_forward", "asg_forward_only", "asg_backward", "asg_full_forward", "asg_full_backward",
              "asg_aligned_forward", "asg_aligned_backward", "asg_state_bytes", "asg_scratch_bytes",
              "asg_ctx_create", "asg_ctx_destroy", "asg_hip_version", "asg_hip_strerror"):
        assert n in names


def test_library_loads_and_exports_every_declared_symbol():
    from torch_asg_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with python torch_asg_amd/csrc/build.py"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert sorted(_lib.SYMBOLS) == _declared_symbols()
    assert _lib.lib().asg_hip_version() == _lib.ABI_VERSION == 230


def test_sizes_and_argument_validation_without_gpu():
    from torch_asg_amd import _lib
    L = _lib.lib()
    p = _lib.AsgProblem()
    p.T, p.B, p.N, p.S, p.dtype = 400, 64, 40, 30, _lib.ASG_DTYPE_F32
    st = L.asg_state_bytes(ctypes.byref(p))
    # saved state is O(T*B*(N+S)): 2 alpha/beta pairs (+ two scalars per frame: the alpha pass's scale log), never the
    # reference's O(T*B*N*N) path_contrib
    assert 2 * 400 * 64 * (40 + 30 + 1) * 4 <= st < 2 * 400 * 64 * (40 + 30 + 1) * 4 * 1.01
    assert L.asg_scratch_bytes(ctypes.byref(p)) >= 40 * 40 * 4
    p.dtype = _lib.ASG_DTYPE_F64
    assert L.asg_state_bytes(ctypes.byref(p)) >= 2 * st * 0.95
    # null problem / null buffers are rejected before anything touches a device
    assert L.asg_forward(None, None, None, 0, None, None, 0, None) == 1
    assert L.asg_forward(None, ctypes.byref(p), None, 0, None, None, 0, None) == 1
    p.dtype = 7
    assert L.asg_full_forward(ctypes.byref(p), None, 0, None, 0, None) == 1
    assert L.asg_hip_strerror(0) == b"ok" and b"invalid" in L.asg_hip_strerror(1)
    assert b"small" in L.asg_hip_strerror(3)


def test_product_package_never_references_the_oracle():
    # the oracle is test infrastructure: nothing under torch_asg_amd/ may import or mention it
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "torch_asg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(d, f)).read()
                if re.search(r"\boracle\b", txt) and "no CPU fallback" not in txt and f != "_lib.py":
                    bad.append(os.path.join(d, f))
                if re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from torch_asg_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_cpp_fast_path_loads_and_declines_cpu_tensors():
    """torch_asg_amd/_binding.so (csrc/binding.cpp) is host plumbing above the C ABI: it must import without a GPU,
    take the entry points by address, and hand anything that is not the plain device case back to the Python path
    (None) -- where CPU tensors raise, there being no CPU implementation."""
    import torch
    from torch_asg_amd import _lib
    from torch_asg_amd import asg as A
    assert os.path.exists(_lib.BINDING_PATH), "build with python torch_asg_amd/csrc/build.py"
    be = A.HipBackend()
    assert be.binding is not None
    x = torch.zeros(5, 2, 4)
    tr = torch.zeros(4, 4)
    tg = torch.zeros(2, 3, dtype=torch.int64)
    assert be.binding.try_loss_forward(x, tr, tg, None, None, 2, 2) is None
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        A.ASGLossFunction.apply(x, tr, tg, None, None, "mean", 2)


def test_cpp_fast_path_can_be_switched_off(monkeypatch):
    from torch_asg_amd import asg as A
    monkeypatch.setenv("ASG_NO_BINDING", "1")
    assert A.HipBackend().binding is None
