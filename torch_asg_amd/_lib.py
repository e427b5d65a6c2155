"""ctypes binding of libasg_hip.so (C ABI: include/asg_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing this module's
`lib()` raises.  The product path never routes through oracle/ or any CPU implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASG_HIP_LIB") or os.path.join(_HERE, "csrc", "libasg_hip.so")

ASG_DTYPE_F32, ASG_DTYPE_F64, ASG_DTYPE_BF16 = 0, 1, 2
FLAG_STREAMS, FLAG_SINGLE_LAUNCH, FLAG_ALPHA_SCORES = 1, 2, 8

# every symbol include/asg_hip.h declares
SYMBOLS = ["asg_hip_version", "asg_hip_strerror", "asg_ctx_create", "asg_ctx_destroy", "asg_stream_capture_id", "asg_state_bytes",
           "asg_scratch_bytes", "asg_full_forward", "asg_full_backward", "asg_aligned_forward",
           "asg_aligned_backward", "asg_forward", "asg_forward_only", "asg_backward", "asg_loss_forward",
           "asg_loss_backward", "asg_viterbi_work_bytes", "asg_viterbi", "asg_loss_fused_supported",
           "asg_loss_fused_scratch_bytes", "asg_loss_fused_sync_bytes", "asg_loss_fused_forward",
           "asg_loss_fused_backward", "asg_cluster_timeouts", "asg_reload_env", "asg_loss_forward_only", "asg_loss_forward_only_scores_bytes"]
ABI_VERSION = 230        # include/asg_hip.h: ASG_HIP_VERSION this package was written against


class AsgProblem(ctypes.Structure):
    _fields_ = [("inputs", ctypes.c_void_p), ("inputs_strides", ctypes.c_int64 * 3),
                ("transition", ctypes.c_void_p), ("transition_strides", ctypes.c_int64 * 2),
                ("targets", ctypes.c_void_p), ("targets_strides", ctypes.c_int64 * 2),
                ("input_lengths", ctypes.c_void_p), ("target_lengths", ctypes.c_void_p),
                ("T", ctypes.c_int64), ("B", ctypes.c_int64), ("N", ctypes.c_int64), ("S", ctypes.c_int64),
                ("dtype", ctypes.c_int32), ("inputs_dtype", ctypes.c_int32)]


_LIB = None


def lib():
    """Load libasg_hip.so once; fail loudly if it is not there."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "torch_asg_amd: %s not found. Build it with `python torch_asg_amd/csrc/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    pp = ctypes.POINTER(AsgProblem)
    L.asg_hip_version.restype = ci
    if int(L.asg_hip_version()) != ABI_VERSION:
        raise RuntimeError("torch_asg_amd: %s reports ABI version %d, this package needs %d -- rebuild it with "
                           "`python torch_asg_amd/csrc/build.py`" % (LIB_PATH, int(L.asg_hip_version()), ABI_VERSION))
    L.asg_cluster_timeouts.restype = ctypes.c_uint
    L.asg_cluster_timeouts.argtypes = []
    L.asg_reload_env.restype = None
    L.asg_reload_env.argtypes = []
    L.asg_hip_strerror.restype = ctypes.c_char_p
    L.asg_hip_strerror.argtypes = [ci]
    L.asg_ctx_create.argtypes = [ctypes.POINTER(vp)]
    L.asg_ctx_destroy.argtypes = [vp]
    L.asg_stream_capture_id.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong)]
    L.asg_state_bytes.restype = sz
    L.asg_state_bytes.argtypes = [pp]
    L.asg_scratch_bytes.restype = sz
    L.asg_scratch_bytes.argtypes = [pp]
    L.asg_full_forward.argtypes = [pp, vp, sz, vp, ci, vp]
    L.asg_aligned_forward.argtypes = [pp, vp, sz, vp, ci, vp]
    L.asg_full_backward.argtypes = [pp, vp, sz, vp, vp, sz, vp, vp, vp]
    L.asg_aligned_backward.argtypes = [pp, vp, sz, vp, vp, sz, vp, vp, vp]
    L.asg_forward.argtypes = [vp, pp, vp, sz, vp, vp, ci, vp]
    L.asg_forward_only.argtypes = [vp, pp, vp, sz, vp, vp, ci, vp]
    L.asg_backward.argtypes = [vp, pp, vp, sz, vp, vp, vp, sz, vp, vp, ci, vp]
    L.asg_loss_forward.argtypes = [vp, pp, vp, sz, ci, vp, vp, ci, vp]
    L.asg_loss_backward.argtypes = [vp, pp, vp, sz, ci, vp, vp, sz, vp, vp, ci, vp]
    L.asg_loss_forward_only_scores_bytes.restype = sz
    L.asg_loss_forward_only_scores_bytes.argtypes = [pp]
    L.asg_loss_forward_only.argtypes = [vp, pp, vp, sz, ci, vp, vp, sz, ci, vp]
    L.asg_viterbi_work_bytes.restype = sz
    L.asg_viterbi_work_bytes.argtypes = [pp]
    L.asg_viterbi.argtypes = [vp, pp, vp, sz, vp, vp, ci, vp]
    L.asg_loss_fused_supported.argtypes = [pp]
    L.asg_loss_fused_scratch_bytes.restype = sz
    L.asg_loss_fused_scratch_bytes.argtypes = [pp]
    L.asg_loss_fused_sync_bytes.restype = sz
    L.asg_loss_fused_sync_bytes.argtypes = [pp]
    L.asg_loss_fused_forward.argtypes = [pp, vp, sz, ci, vp, vp, vp, sz, vp, vp, ci, vp]
    L.asg_loss_fused_backward.argtypes = [pp, vp, sz, ci, vp, vp, sz, vp, vp, ci, vp]
    for name in SYMBOLS:
        getattr(L, name)
    _LIB = L
    return L


def check(status, what):
    if status != 0:
        msg = lib().asg_hip_strerror(int(status)).decode()
        raise RuntimeError("torch_asg_amd: %s failed: %s (status %d)" % (what, msg, status))


# what the C++ fast path of ASGLossFunction (csrc/binding.cpp) calls, in the order its init() expects
BINDING_SYMBOLS = ["asg_state_bytes", "asg_scratch_bytes", "asg_loss_fused_scratch_bytes", "asg_loss_fused_sync_bytes",
                   "asg_loss_fused_supported", "asg_stream_capture_id", "asg_hip_strerror", "asg_loss_forward",
                   "asg_loss_backward", "asg_loss_fused_forward", "asg_loss_fused_backward", "asg_cluster_timeouts",
                   "asg_loss_forward_only"]
BINDING_PATH = os.path.join(_HERE, "_binding.so")


def binding(backend):
    """The C++ host fast path (torch_asg_amd/_binding.so), one object per backend, initialised with the entry points of the library that
    `lib()` loaded (so ASG_HIP_LIB variants are honoured), or None when it has not been built or ASG_NO_BINDING=1 --
    the Python statements in asg.py then do the same work, only slower (host time)."""
    if os.environ.get("ASG_NO_BINDING", "0") not in ("", "0") or not os.path.exists(BINDING_PATH):
        return None
    L = lib()
    try:
        # a stale or ABI-mismatched _binding.so (torch upgraded, other C++ ABI, missing libtorch_hip) must not take the package
        # down: the Python statements do the same work
        from . import _binding
        return _binding.Fast([ctypes.cast(getattr(L, n), ctypes.c_void_p).value for n in BINDING_SYMBOLS], backend)
    except (ImportError, OSError, RuntimeError, AttributeError) as e:
        import warnings
        warnings.warn("torch_asg_amd: the C++ host fast path (_binding.so) is unusable (%s: %s); using the Python path -- rebuild "
                      "with `python torch_asg_amd/csrc/build.py --binding`" % (type(e).__name__, e))
        return None
