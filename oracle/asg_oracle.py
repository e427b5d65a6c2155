"""oracle/asg_oracle.py -- ctypes/numpy front-end of the CPU oracle. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; torch_asg_amd/ never does (the product path fails loudly without its HIP library).

`asg_loss()` restates the host logic of the reference's ASGLoss.forward
(/root/reference/torch_asg/asg.py:109-142) and the autograd chain behind it
(asg.py:7-55): defaults for missing lengths, truncation of targets longer than T,
loss = full - aligned, reduction, and the gradient of the reduced loss.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libasg_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("asg_oracle.c", "asg_oracle_impl.inc")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libasg_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64 only, got %s" % dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _istr(a):
    return (ctypes.c_int64 * 3)(*[s // a.itemsize for s in a.strides])


def _lens(x, B):
    if x is None:
        return None
    x = np.ascontiguousarray(np.asarray(x), dtype=np.int64)
    assert x.shape == (B,)
    return x


def full_forward(inputs, transition, input_lengths=None):
    """-> scores[B], alpha[T,B,N], beta[T,B,N]   (fully_connected_lattice.cpp:65-91)"""
    inputs = np.asarray(inputs)
    T, B, N = inputs.shape
    sfx = _sfx(inputs.dtype)
    tr = np.ascontiguousarray(transition, dtype=inputs.dtype)
    il = _lens(input_lengths, B)
    scores = np.empty(B, inputs.dtype)
    alpha = np.empty((T, B, N), inputs.dtype)
    beta = np.empty((T, B, N), inputs.dtype)
    rc = getattr(lib(), "asg_oracle_full_forward_" + sfx)(
        _p(inputs), _istr(inputs), _p(tr), _p(il), ctypes.c_int64(T), ctypes.c_int64(B), ctypes.c_int64(N),
        _p(scores), _p(alpha), _p(beta))
    assert rc == 0, rc
    return scores, alpha, beta


def full_backward(grad_out, alpha, beta, inputs, transition):
    """-> grad_transition[N,N], grad_inputs[T,B,N]   (fully_connected_lattice.cpp:93-105)"""
    inputs = np.asarray(inputs)
    T, B, N = inputs.shape
    sfx = _sfx(inputs.dtype)
    tr = np.ascontiguousarray(transition, dtype=inputs.dtype)
    g = np.ascontiguousarray(grad_out, dtype=inputs.dtype)
    gtr = np.empty((N, N), inputs.dtype)
    gin = np.empty((T, B, N), inputs.dtype)
    rc = getattr(lib(), "asg_oracle_full_backward_" + sfx)(
        _p(g), _p(np.ascontiguousarray(alpha)), _p(np.ascontiguousarray(beta)), _p(inputs), _istr(inputs), _p(tr),
        ctypes.c_int64(T), ctypes.c_int64(B), ctypes.c_int64(N), _p(gtr), _p(gin))
    assert rc == 0, rc
    return gtr, gin


def aligned_forward(inputs, targets, transition, input_lengths=None, target_lengths=None):
    """-> scores[B], alpha[T,B,S], beta[T,B,S]   (force_aligned_lattice.cpp:266-319)"""
    inputs = np.asarray(inputs)
    T, B, N = inputs.shape
    tg = np.ascontiguousarray(targets, dtype=np.int64)
    S = tg.shape[1]
    sfx = _sfx(inputs.dtype)
    tr = np.ascontiguousarray(transition, dtype=inputs.dtype)
    il, tl = _lens(input_lengths, B), _lens(target_lengths, B)
    scores = np.empty(B, inputs.dtype)
    alpha = np.empty((T, B, S), inputs.dtype)
    beta = np.empty((T, B, S), inputs.dtype)
    rc = getattr(lib(), "asg_oracle_aligned_forward_" + sfx)(
        _p(inputs), _istr(inputs), _p(tg), _p(tr), _p(il), _p(tl),
        ctypes.c_int64(T), ctypes.c_int64(B), ctypes.c_int64(N), ctypes.c_int64(S),
        _p(scores), _p(alpha), _p(beta))
    assert rc == 0, rc
    return scores, alpha, beta


def aligned_backward(grad_out, alpha, beta, targets, transition, input_lengths, target_lengths, num_labels):
    """-> grad_transition[N,N], grad_inputs[T,B,N]   (force_aligned_lattice.cpp:321-356)"""
    alpha = np.ascontiguousarray(alpha)
    T, B, S = alpha.shape
    N = num_labels
    sfx = _sfx(alpha.dtype)
    tg = np.ascontiguousarray(targets, dtype=np.int64)
    tr = np.ascontiguousarray(transition, dtype=alpha.dtype)
    g = np.ascontiguousarray(grad_out, dtype=alpha.dtype)
    il, tl = _lens(input_lengths, B), _lens(target_lengths, B)
    gtr = np.empty((N, N), alpha.dtype)
    gin = np.empty((T, B, N), alpha.dtype)
    rc = getattr(lib(), "asg_oracle_aligned_backward_" + sfx)(
        _p(g), _p(alpha), _p(np.ascontiguousarray(beta)), _p(tg), _p(tr), _p(il), _p(tl),
        ctypes.c_int64(T), ctypes.c_int64(B), ctypes.c_int64(N), ctypes.c_int64(S), _p(gtr), _p(gin))
    assert rc == 0, rc
    return gtr, gin


def viterbi(inputs, targets, transition, input_lengths=None, target_lengths=None):
    """Best-path force alignment (max-plus version of force_aligned_lattice.cpp:84-111; not in the reference).
    -> scores[B], path[B,T] int64 (target position per frame, -1 outside the utterance / infeasible)"""
    inputs = np.asarray(inputs)
    T, B, N = inputs.shape
    tg = np.ascontiguousarray(targets, dtype=np.int64)
    S = tg.shape[1]
    sfx = _sfx(inputs.dtype)
    tr = np.ascontiguousarray(transition, dtype=inputs.dtype)
    il, tl = _lens(input_lengths, B), _lens(target_lengths, B)
    scores = np.empty(B, inputs.dtype)
    path = np.empty((B, T), np.int64)
    rc = getattr(lib(), "asg_oracle_viterbi_" + sfx)(
        _p(inputs), _istr(inputs), _p(tg), _p(tr), _p(il), _p(tl),
        ctypes.c_int64(T), ctypes.c_int64(B), ctypes.c_int64(N), ctypes.c_int64(S), _p(scores), _p(path))
    assert rc == 0, rc
    return scores, path


def brute_force_viterbi(inputs, targets, transition, input_length, target_length):
    """Exhaustive best alignment for ONE tiny utterance (pure Python). inputs [T,N] float64.
    Returns (best_score, best_position_path) with ties resolved like the recursion (prefer staying,
    i.e. the lexicographically LARGEST sequence of advance frames... see tests: only scores are compared
    when several paths tie)."""
    import itertools
    T = int(input_length)
    tgt = [int(x) for x in targets[:int(target_length)]]
    L = len(tgt)
    if L < 1 or L > T:
        return -np.inf, None
    best, bestp = -np.inf, None
    for cuts in itertools.combinations(range(1, T), L - 1):
        seg, k = [0] * T, 0
        for t in range(T):
            if k < L - 1 and t == cuts[k]:
                k += 1
            seg[t] = k
        lab = [tgt[k] for k in seg]
        sc = inputs[0, lab[0]]
        for t in range(1, T):
            sc += transition[lab[t], lab[t - 1]] + inputs[t, lab[t]]
        if sc > best:
            best, bestp = sc, seg
    return best, bestp


def asg_loss(inputs, targets, transition, input_lengths=None, target_lengths=None,
             reduction="mean", grad_out=None, need_grad=True):
    """Whole ASGLoss forward+backward on the CPU oracle.

    Restates /root/reference/torch_asg/asg.py:109-142 (defaults :113-117, truncation
    :119-122, full - aligned :128, reduction :137-142).  `grad_out` is the gradient
    flowing into the *reduced* loss (scalar for mean/sum, [B] for 'none'); default 1.
    Returns a dict of numpy arrays.
    """
    inputs = np.asarray(inputs)
    T, B, N = inputs.shape
    targets = np.asarray(targets)
    S = targets.shape[1]
    tl = np.full(B, S, np.int64) if target_lengths is None else np.asarray(target_lengths, np.int64)
    il = np.full(B, T, np.int64) if input_lengths is None else np.asarray(input_lengths, np.int64)
    if S > T:
        S = T
        targets = targets[:, :S]
        tl = np.minimum(tl, S)
    full, fa, fb = full_forward(inputs, transition, il)
    ali, aa, ab = aligned_forward(inputs, targets, transition, il, tl)
    with np.errstate(invalid="ignore"):
        per_utt = full - ali
    if reduction == "sum":
        loss = per_utt.sum()
    elif reduction == "mean":
        loss = per_utt.mean()
    else:
        loss = per_utt
    out = {"loss": loss, "loss_per_utt": per_utt, "full_scores": full, "aligned_scores": ali}
    if need_grad:
        if reduction == "none":
            g = np.ones(B, inputs.dtype) if grad_out is None else np.asarray(grad_out, inputs.dtype)
        else:
            g0 = 1.0 if grad_out is None else float(grad_out)
            g = np.full(B, g0 / B if reduction == "mean" else g0, inputs.dtype)
        gtr_f, gin_f = full_backward(g, fa, fb, inputs, transition)
        gtr_a, gin_a = aligned_backward(-g, aa, ab, targets, transition, il, tl, N)
        out["grad_inputs"] = gin_f + gin_a
        out["grad_transition"] = gtr_f + gtr_a
        out["grad_inputs_full"], out["grad_inputs_aligned"] = gin_f, gin_a
        out["grad_transition_full"], out["grad_transition_aligned"] = gtr_f, gtr_a
    return out


def brute_force_scores(inputs, targets, transition, input_length, target_length):
    """Path enumeration for ONE tiny utterance (pure Python; T<=6, N<=4).

    Independent of the recursions: sums exp(score) over every label path (full) and
    over every path that collapses to the target with >=1 frame per target position
    (aligned).  inputs [T,N] float64.  Returns (S_full, S_aligned).
    """
    import itertools
    import math
    T, N = inputs.shape
    T = int(input_length)
    tgt = [int(x) for x in targets[:int(target_length)]]

    def path_score(p):
        s = inputs[0, p[0]]
        for t in range(1, T):
            s += transition[p[t], p[t - 1]] + inputs[t, p[t]]
        return s

    full = [path_score(p) for p in itertools.product(range(N), repeat=T)]
    ali = []
    L = len(tgt)
    if 1 <= L <= T:
        # choose segment boundaries: positions where the alignment advances
        for cuts in itertools.combinations(range(1, T), L - 1):
            seg = [0] * T
            k = 0
            for t in range(T):
                if k < L - 1 and t == cuts[k]:
                    k += 1
                seg[t] = k
            ali.append(path_score([tgt[k] for k in seg]))

    def lse(v):
        if not v:
            return -math.inf
        m = max(v)
        return m + math.log(sum(math.exp(x - m) for x in v))

    return lse(full), lse(ali)
