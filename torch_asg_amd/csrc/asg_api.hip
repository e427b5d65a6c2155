// torch_asg_amd/csrc/asg_api.hip -- the C ABI declared in include/asg_hip.h.
// Argument validation, state/scratch layout, path selection (small / generic), stream fork-join.
// No torch, no pybind: plain HIP runtime calls only.
#include "../../include/asg_hip.h"
#include "asg_kernels.h"
#include "asg_common.h"

#include <hip/hip_runtime.h>
#include <new>
#include <atomic>
#include <cstdlib>

using namespace asg;

namespace asg {
size_t bwd_scratch_bytes_small(int elem, int T, int B, int N, int S, int *chunk, int *nchunks) {
    // at most ~512 workgroups: the assembly kernel holds two workgroups per compute unit (243 VGPRs), so 512 are ONE round on 256
    // CUs; anything above it is a second, mostly empty round (round 3 rounded UP: B = 192 got 3 x 192 = 576 workgroups and ran
    // slower than B = 256 with 2 x 256).  At least 16 frames per workgroup.
    int target = 512;
#ifdef ASG_DEV_PROBES
    if (const char *ev = getenv("ASG_BWD_WGS")) target = atoi(ev) > 0 ? atoi(ev) : target;   // developer probe
#endif
    int nch = target / B;
    if (nch < 1) nch = 1;
    int ch = (T + nch - 1) / nch;
    if (ch < 16) ch = 16;
    ch = (ch + 15) / 16 * 16;      // whole 16-frame blocks (bwd_mfma_kernel)
    nch = (T + ch - 1) / ch;
    if (nch < 1) nch = 1;
    if (chunk) *chunk = ch;
    if (nchunks) *nchunks = nch;
    return (size_t) B * nch * N * N * elem;
}
}  // namespace asg

namespace asg {
namespace {
std::atomic<const Knobs *> g_knobs{nullptr};
int env_int(const char *name) { const char *v = getenv(name); return v ? atoi(v) : -1; }
const Knobs *read_knobs() {
    Knobs *k = new Knobs();          // (a handful per process at most: never freed, readers may still hold the old one)
    k->fork_in_capture = env_int("ASG_FORK_IN_CAPTURE");
    k->pair_min_b = env_int("ASG_PAIR_MIN_B");
    k->bwd_rowsum = env_int("ASG_BWD_ROWSUM");
    k->no_cluster = env_int("ASG_NO_CLUSTER");
    k->no_mid = env_int("ASG_NO_MID");
    k->no_tile_step = env_int("ASG_NO_TILE_STEP");
    k->step_one_tile = env_int("ASG_STEP_ONE_TILE");
    k->step_row_blocks = env_int("ASG_STEP_ROW_BLOCKS");
    k->step_full_tile = env_int("ASG_STEP_FULL_TILE");
    k->step_no_bf3 = env_int("ASG_STEP_NO_BF3");
    k->step_bf3_min_b = env_int("ASG_STEP_BF3_MIN_B");
    const char *ak = getenv("ASG_ALIGNED_KERNEL");
    k->aligned_kernel = ak ? ak[0] : 0;
    return k;
}
}  // namespace
const Knobs &knobs() {
    const Knobs *k = g_knobs.load(std::memory_order_acquire);
    if (!k) {
        const Knobs *fresh = read_knobs();
        if (g_knobs.compare_exchange_strong(k, fresh, std::memory_order_acq_rel)) k = fresh;
        else delete fresh;
    }
    return *k;
}
}  // namespace asg

// Host-side handles only (side stream + fork/join events of ASG_FLAG_STREAMS); no device memory, no per-call state:
// everything a call mutates on the device lives in the caller-owned `state` buffer of that call.
struct asg_ctx {
    hipStream_t side;
    hipEvent_t fork, join;
    int device;
};

namespace {

inline int hip_status(hipError_t e) { return e == hipSuccess ? ASG_OK : ASG_ERR_HIP_BASE + (int) e; }

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t) 255; }

struct Layout {
    size_t ah, bh, ab, bb, klog, ehat, fhat, rmax, cmax, etile, ftile, asu, asi, dbg, ticket, work, total;
    int npad;
};

inline bool small_full(int64_t N) { return N <= 64; }
inline bool small_aligned(int64_t S) { return S <= 64; }

Layout make_layout(const asg_problem *p) {
    Layout L{};
    const size_t e = p->dtype == ASG_DTYPE_F64 ? 8 : 4;
    const size_t T = p->T, B = p->B, N = p->N, S = p->S < 1 ? 1 : p->S;
    size_t off = 0;
    L.ah = off; off = align_up(off + B * T * N * e);
    L.bh = off; off = align_up(off + B * T * N * e);
    // aligned states: the problem's type on the small path, doubles from the long-target kernels (S > 64: asg_generic.hip, AlignedState)
    const size_t ea = small_aligned((int64_t) S) ? e : 8;
    L.ab = off; off = align_up(off + B * T * S * ea);
    L.bb = off; off = align_up(off + B * T * S * ea);
    if (small_full(p->N)) { L.klog = off; off = align_up(off + B * T * 2 * e); }      // scale log of the alpha pass (ScaleLog)
    L.npad = small_full(p->N) ? (int) ((N + 7) / 8 * 8) : (int) ((N + 3) / 4 * 4);
    L.ehat = off; off = align_up(off + N * L.npad * e);
    L.rmax = off; off = align_up(off + N * e);
    if (!small_full(p->N)) {
        L.fhat = off; off = align_up(off + N * L.npad * e);
        L.cmax = off; off = align_up(off + N * e);
        const size_t tb = step_tile_bytes_generic((int) e, (int) N, (int) B);
        L.etile = off; off = align_up(off + tb);
        L.ftile = off; off = align_up(off + tb);
    }
    L.asu = off; off = align_up(off + B * S * 2 * e);
    L.asi = off; off = align_up(off + B * S * 2 * sizeof(int));
    L.dbg = off; off = align_up(off + 512);      // developer timing stamps (ASG_PROBE builds only)
    L.ticket = off; off = align_up(off + 256);   // arrival ticket of the in-kernel loss reduction (zeroed per call)
    if (!small_full(p->N)) { L.work = off; off = align_up(off + fwd_work_bytes_generic((int) e, (int) T, (int) B, (int) N)); }
    L.total = off;
    return L;
}

int check_problem(const asg_problem *p, bool need_targets, bool allow_bf16 = false) {
    if (!p) return ASG_ERR_INVALID;
    if (p->dtype != ASG_DTYPE_F32 && p->dtype != ASG_DTYPE_F64) return ASG_ERR_INVALID;
    if (p->inputs_dtype != 0 && !(p->inputs_dtype == ASG_DTYPE_BF16 && p->dtype == ASG_DTYPE_F32)) return ASG_ERR_INVALID;
    if (p->inputs_dtype != 0 && !allow_bf16) return ASG_ERR_UNSUPPORTED;      // bfloat16 emissions: the fused pair only
    if (p->T < 1 || p->B < 1 || p->N < 1) return ASG_ERR_INVALID;
    if (!p->inputs || !p->transition) return ASG_ERR_INVALID;
    if (need_targets && (p->S < 1 || !p->targets)) return ASG_ERR_INVALID;
    if (p->T > (1 << 30) || p->B > (1 << 30) || p->N > (1 << 30) || p->S > (1 << 30)) return ASG_ERR_UNSUPPORTED;
    if (p->S > 8192) return ASG_ERR_UNSUPPORTED;              // aligned kernels: at most eight target positions per thread of a workgroup, a frame's states in LDS
    if (p->S > 1024 && (double) p->N * (p->dtype == ASG_DTYPE_F64 ? 8.0 : 4.0) > 150.0 * 1024.0) return ASG_ERR_UNSUPPORTED;   // ... and beyond 1024 the label scatter is one LDS row of N words (bwd_aligned_strip_kernel)
    {
        // the kernels address state and gradient rows through 32-bit buffer offsets
        const double e = p->dtype == ASG_DTYPE_F64 ? 8.0 : 4.0;
        const double w = (double) (p->N > p->S ? p->N : p->S);
        if ((double) p->T * (double) p->B * w * e >= 4294967296.0 && small_full(p->N)) return ASG_ERR_UNSUPPORTED;
    }
    return ASG_OK;
}

Problem to_problem(const asg_problem *p) {
    Problem P{};
    P.inputs = p->inputs;
    P.is0 = p->inputs_strides[0]; P.is1 = p->inputs_strides[1]; P.is2 = p->inputs_strides[2];
    P.transition = p->transition;
    P.ts0 = p->transition_strides[0]; P.ts1 = p->transition_strides[1];
    P.targets = p->targets;
    P.gs0 = p->targets_strides[0]; P.gs1 = p->targets_strides[1];
    P.in_len = p->input_lengths;
    P.tg_len = p->target_lengths;
    P.T = (int) p->T; P.B = (int) p->B; P.N = (int) p->N; P.S = (int) (p->S < 1 ? 1 : p->S);
    P.in_bf16 = p->inputs_dtype == ASG_DTYPE_BF16 ? 1 : 0;
    return P;
}

State to_state(const asg_problem *p, const void *state) {
    Layout L = make_layout(p);
    char *base = (char *) state;
    State W{};
    if (base) {
        W.ah = base + L.ah; W.bh = base + L.bh; W.ab = base + L.ab; W.bb = base + L.bb;
        W.ehat = base + L.ehat; W.rmax = base + L.rmax;
        if (!small_full(p->N)) { W.fhat = base + L.fhat; W.cmax = base + L.cmax; W.etile = base + L.etile; W.ftile = base + L.ftile; }
        W.asu = base + L.asu; W.asi = (int *) (base + L.asi);
        W.dbg = base + L.dbg;
        W.ticket = (unsigned *) (base + L.ticket);
        if (!small_full(p->N)) W.work = base + L.work;
        else W.klog = base + L.klog;
    }
    W.npad = L.npad;
    return W;
}

// Is `stream` being captured into a hipGraph?  (ASG_FORK_IN_CAPTURE=1 / 0: developer override of what run_forward does with it.)
inline bool capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void) hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}

template <typename R>
int run_forward(asg_ctx *ctx, const asg_problem *p, void *state, void *full_scores, void *aligned_scores,
                int mask, bool store, int flags, hipStream_t stream, void *loss = nullptr, int reduction = 0,
                unsigned *ticket = nullptr) {
    Problem P = to_problem(p);
    // Two streams (fork / join through ctx) only help when the two lattices' kernels really run side by side.  Recorded into a
    // hipGraph, ROCm 7.2 replays the two branches of the SHORT kernels one after the other with ~12 us per cross-queue edge
    // (profiles/r04_streams_graph_trace.txt: cfg 3, fwd_duo_kernel 48.6 us, then a 13 us gap, then the aligned chains 37 us,
    // then 11 us before the backward launch): while capturing, the small path records its passes on the caller's stream.
    bool fork_ok = true;
    {
        const int mode = knobs().fork_in_capture;        // -1: default policy
        if (mode == 0) fork_ok = !capturing(stream);
        else if (mode < 0) fork_ok = !(small_full(p->N) && small_aligned(p->S) && capturing(stream));
    }
    State W = to_state(p, state);
    FwdOut O{};
    O.full_scores = full_scores;
    O.aligned_scores = aligned_scores;
    if (loss && small_full(p->N) && small_aligned(p->S)) {
        // the last beta pass to finish reduces the loss; its arrival ticket is part of THIS call's state buffer
        // and is zeroed on the launch stream ahead of the kernels (a memset node under graph capture)
        O.loss = loss;
        O.counter = ticket ? ticket : W.ticket;           // (the evaluation route has no state buffer: its caller lends 256 bytes)
        O.reduction = reduction;
        O.expected = 2 * (int) p->B;
        // (a kernel, not hipMemsetAsync: asg_common.h::zero_async says why)
        hipError_t me = zero_async(O.counter, 256, stream);
        if (me != hipSuccess) return hip_status(me);
    }
    if ((flags & ASG_FLAG_ALPHA_SCORES) && store) {
        if (full_scores) O.full_scores_alpha = (R *) full_scores + p->B;
        if (aligned_scores) O.aligned_scores_alpha = (R *) aligned_scores + p->B;
    }
#ifdef ASG_DEV_PROBES
    if (const char *dm = getenv("ASG_DEBUG_MASK")) mask &= atoi(dm);      // developer probe: time single passes
#endif
    const int full_mask = mask & (kFullAlpha | kFullBeta);
    const int ali_mask = mask & (kAlignedAlpha | kAlignedBeta);
    const bool sf = small_full(p->N), sa = small_aligned(p->S);
    if ((full_mask && !sf) || (ali_mask && !sa)) {
        // generic path (any N / S up to 1024)
        hipError_t e = hipSuccess;
        if (full_mask && !sf) {
            e = launch_prep_generic<R>(P, W, stream);
            if (e != hipSuccess) return hip_status(e);
        }
        // two streams also for the default launch mode: on this route the two lattices are separate launches anyway (one
        // launch per lattice, or per frame), and overlapping them is worth more than the fork / join costs (T=1000 B=64
        // N=40 S=200: 492 -> 420 us per step inside a hipGraph)
        bool two = (flags & (ASG_FLAG_STREAMS | ASG_FLAG_SINGLE_LAUNCH)) && ctx && full_mask && ali_mask && fork_ok;
        hipStream_t s2 = two ? ctx->side : stream;
        if (two) {
            if ((e = hipEventRecord(ctx->fork, stream)) != hipSuccess) return hip_status(e);
            if ((e = hipStreamWaitEvent(s2, ctx->fork, 0)) != hipSuccess) return hip_status(e);
        }
        if (full_mask) {
            e = sf ? launch_fwd_small<R>(P, W, O, full_mask, store, stream)
                   : launch_fwd_generic<R>(P, W, O, full_mask, store, stream);
            if (e != hipSuccess) return hip_status(e);
        }
        if (ali_mask) {
            e = sa ? launch_fwd_small<R>(P, W, O, ali_mask, store, s2)
                   : launch_fwd_generic<R>(P, W, O, ali_mask, store, s2);
            if (e != hipSuccess) return hip_status(e);
        }
        if (two) {
            if ((e = hipEventRecord(ctx->join, s2)) != hipSuccess) return hip_status(e);
            if ((e = hipStreamWaitEvent(stream, ctx->join, 0)) != hipSuccess) return hip_status(e);
        }
        return ASG_OK;
    }
    hipError_t e;
    if ((flags & ASG_FLAG_STREAMS) && ctx && full_mask && ali_mask && fork_ok) {
        // fork: aligned passes on the side stream, full passes on the caller's stream; join back.
        if ((e = hipEventRecord(ctx->fork, stream)) != hipSuccess) return hip_status(e);
        if ((e = hipStreamWaitEvent(ctx->side, ctx->fork, 0)) != hipSuccess) return hip_status(e);
        if ((e = launch_fwd_small<R>(P, W, O, full_mask, store, stream)) != hipSuccess) return hip_status(e);
        if ((e = launch_fwd_small<R>(P, W, O, ali_mask, store, ctx->side)) != hipSuccess) return hip_status(e);
        if ((e = hipEventRecord(ctx->join, ctx->side)) != hipSuccess) return hip_status(e);
        if ((e = hipStreamWaitEvent(stream, ctx->join, 0)) != hipSuccess) return hip_status(e);
        return ASG_OK;
    }
    if (flags & ASG_FLAG_SINGLE_LAUNCH) return hip_status(launch_fwd_small<R>(P, W, O, mask, store, stream));
    if (full_mask && (e = launch_fwd_small<R>(P, W, O, full_mask, store, stream)) != hipSuccess) return hip_status(e);
    if (ali_mask && (e = launch_fwd_small<R>(P, W, O, ali_mask, store, stream)) != hipSuccess) return hip_status(e);
    return ASG_OK;
}

template <typename R>
int run_backward(const asg_problem *p, const void *state, const void *grad_full, const void *grad_aligned,
                 void *scratch, size_t scratch_bytes, void *grad_transition, void *grad_inputs, int parts,
                 hipStream_t stream, int gstride = 1, double gscale = 1.0, int neg_aligned = 0) {
    Problem P = to_problem(p);
    State W = to_state(p, state);
    BwdArgs A{};
    A.grad_full = grad_full ? grad_full : grad_aligned;
    A.grad_aligned = grad_aligned;
    A.gstride = gstride;
    A.gscale = gscale;
    A.neg_aligned = neg_aligned;
    A.grad_inputs = grad_inputs;
    A.grad_transition = grad_transition;
    A.scratch = scratch;
    if (scratch_bytes < asg_scratch_bytes(p)) return ASG_ERR_WORKSPACE;
    const bool sf = small_full(p->N), sa = small_aligned(p->S);
    bwd_scratch_bytes_small((int) sizeof(R), P.T, P.B, P.N, P.S, &A.chunk, &A.nchunks);
    if (sf && (sa || !(parts & 2))) return hip_status(launch_bwd_small<R>(P, W, A, parts, stream));
    // mixed / generic: full-lattice part by whichever path owns N, aligned part by the generic kernels
    hipError_t e = hipSuccess;
    int gparts = 0;
    if (parts & 1) {
        if (sf) {
            if ((e = launch_bwd_small<R>(P, W, A, 1, stream)) != hipSuccess) return hip_status(e);
            gparts |= 4;
        } else {
            gparts |= 1;
        }
    }
    if (parts & 2) gparts |= 2;
    return hip_status(launch_bwd_generic<R>(P, W, A, gparts, stream));
}

}  // namespace

extern "C" {

int asg_hip_version(void) { return ASG_HIP_VERSION; }

const char *asg_hip_strerror(int status) {
    switch (status) {
        case ASG_OK: return "ok";
        case ASG_ERR_INVALID: return "invalid argument";
        case ASG_ERR_UNSUPPORTED: return "unsupported problem shape";
        case ASG_ERR_WORKSPACE: return "state or scratch buffer too small";
        default: break;
    }
    if (status >= ASG_ERR_HIP_BASE) return hipGetErrorString((hipError_t) (status - ASG_ERR_HIP_BASE));
    return "unknown error";
}

unsigned asg_cluster_timeouts(void) { return cluster_timeouts(); }

void asg_reload_env(void) { g_knobs.store(read_knobs(), std::memory_order_release); }

int asg_ctx_create(asg_ctx **out) {
    if (!out) return ASG_ERR_INVALID;
    asg_ctx *c = new (std::nothrow) asg_ctx();
    if (!c) return ASG_ERR_INVALID;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming);
    if (e != hipSuccess) { delete c; return hip_status(e); }
    *out = c;
    return ASG_OK;
}

int asg_stream_capture_id(void *stream, unsigned long long *id) {
    if (!id) return ASG_ERR_INVALID;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    hipError_t e = hipStreamGetCaptureInfo((hipStream_t) stream, &st, &cid);
    if (e != hipSuccess) return hip_status(e);
    *id = st == hipStreamCaptureStatusActive ? (cid ? cid : ~0ull) : 0;
    return ASG_OK;
}

int asg_ctx_destroy(asg_ctx *c) {
    if (!c) return ASG_OK;
    (void) hipEventDestroy(c->fork);
    (void) hipEventDestroy(c->join);
    (void) hipStreamDestroy(c->side);
    delete c;
    return ASG_OK;
}

size_t asg_state_bytes(const asg_problem *p) {
    if (!p || p->T < 1 || p->B < 1 || p->N < 1) return 0;
    asg_problem q = *p;
    if (q.S < 1) q.S = 1;
    return make_layout(&q).total;
}

size_t asg_loss_forward_only_scores_bytes(const asg_problem *p) {
    if (!p || p->B < 1) return 0;
    return align_up(2 * (size_t) p->B * (p->dtype == ASG_DTYPE_F64 ? 8 : 4)) + 256;
}

size_t asg_scratch_bytes(const asg_problem *p) {
    if (!p || p->T < 1 || p->B < 1 || p->N < 1) return 0;
    const int e = p->dtype == ASG_DTYPE_F64 ? 8 : 4;
    const int S = (int) (p->S < 1 ? 1 : p->S);
    size_t a = 256;
    if (small_full(p->N)) {
        size_t s = bwd_scratch_bytes_small(e, (int) p->T, (int) p->B, (int) p->N, S, nullptr, nullptr);
        if (s > a) a = s;
    }
    if (!small_full(p->N) || !small_aligned(S)) {
        size_t s = bwd_scratch_bytes_generic(e, (int) p->T, (int) p->B, (int) p->N, S);
        if (s > a) a = s;
    }
    return a;
}

#define ASG_DISPATCH(p, call_f32, call_f64) ((p)->dtype == ASG_DTYPE_F32 ? (call_f32) : (call_f64))

int asg_full_forward(const asg_problem *p, void *state, size_t state_bytes, void *scores, int flags, void *stream) {
    int rc = check_problem(p, false);
    if (rc) return rc;
    if (!state || !scores) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    flags &= ~(ASG_FLAG_STREAMS | ASG_FLAG_SINGLE_LAUNCH);
    return ASG_DISPATCH(p,
        run_forward<float>(nullptr, p, state, scores, nullptr, kFullAlpha | kFullBeta, true, flags, (hipStream_t) stream),
        run_forward<double>(nullptr, p, state, scores, nullptr, kFullAlpha | kFullBeta, true, flags, (hipStream_t) stream));
}

int asg_aligned_forward(const asg_problem *p, void *state, size_t state_bytes, void *scores, int flags, void *stream) {
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (!state || !scores) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    flags &= ~(ASG_FLAG_STREAMS | ASG_FLAG_SINGLE_LAUNCH);
    return ASG_DISPATCH(p,
        run_forward<float>(nullptr, p, state, nullptr, scores, kAlignedAlpha | kAlignedBeta, true, flags, (hipStream_t) stream),
        run_forward<double>(nullptr, p, state, nullptr, scores, kAlignedAlpha | kAlignedBeta, true, flags, (hipStream_t) stream));
}

int asg_full_backward(const asg_problem *p, const void *state, size_t state_bytes, const void *grad_out,
                      void *scratch, size_t scratch_bytes, void *grad_transition, void *grad_inputs, void *stream) {
    int rc = check_problem(p, false);
    if (rc) return rc;
    if (!state || !grad_out || !scratch || !grad_transition || !grad_inputs) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    return ASG_DISPATCH(p,
        run_backward<float>(p, state, grad_out, nullptr, scratch, scratch_bytes, grad_transition, grad_inputs, 1, (hipStream_t) stream),
        run_backward<double>(p, state, grad_out, nullptr, scratch, scratch_bytes, grad_transition, grad_inputs, 1, (hipStream_t) stream));
}

int asg_aligned_backward(const asg_problem *p, const void *state, size_t state_bytes, const void *grad_out,
                         void *scratch, size_t scratch_bytes, void *grad_transition, void *grad_inputs, void *stream) {
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (!state || !grad_out || !scratch || !grad_transition || !grad_inputs) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    return ASG_DISPATCH(p,
        run_backward<float>(p, state, nullptr, grad_out, scratch, scratch_bytes, grad_transition, grad_inputs, 2, (hipStream_t) stream),
        run_backward<double>(p, state, nullptr, grad_out, scratch, scratch_bytes, grad_transition, grad_inputs, 2, (hipStream_t) stream));
}

int asg_forward(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes,
                void *full_scores, void *aligned_scores, int flags, void *stream) {
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (!state || !full_scores || !aligned_scores) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    return ASG_DISPATCH(p,
        run_forward<float>(ctx, p, state, full_scores, aligned_scores, 15, true, flags, (hipStream_t) stream),
        run_forward<double>(ctx, p, state, full_scores, aligned_scores, 15, true, flags, (hipStream_t) stream));
}

int asg_forward_only(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes,
                     void *full_scores, void *aligned_scores, int flags, void *stream) {
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (!full_scores || !aligned_scores) return ASG_ERR_INVALID;
    if (!small_full(p->N) || !small_aligned(p->S)) {
        if (!state) return ASG_ERR_INVALID;
        if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    } else {
        state = nullptr;
    }
    flags &= ~ASG_FLAG_ALPHA_SCORES;
    return ASG_DISPATCH(p,
        run_forward<float>(ctx, p, state, full_scores, aligned_scores, kFullBeta | kAlignedBeta, false, flags, (hipStream_t) stream),
        run_forward<double>(ctx, p, state, full_scores, aligned_scores, kFullBeta | kAlignedBeta, false, flags, (hipStream_t) stream));
}

size_t asg_viterbi_work_bytes(const asg_problem *p) {
    if (check_problem(p, true) != ASG_OK) return 0;
    return (size_t) p->B * (size_t) p->T * (size_t) ((p->S + 63) / 64) * sizeof(unsigned long long);
}

int asg_viterbi(asg_ctx *ctx, const asg_problem *p, void *work, size_t work_bytes, void *scores, int64_t *path,
                int flags, void *stream) {
    (void) ctx; (void) flags;
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (!work || !scores || !path) return ASG_ERR_INVALID;
    if (work_bytes < asg_viterbi_work_bytes(p)) return ASG_ERR_WORKSPACE;
    const Problem P = to_problem(p);
    return hip_status(ASG_DISPATCH(p, launch_viterbi_small<float>(P, work, scores, path, (hipStream_t) stream),
                                   launch_viterbi_small<double>(P, work, scores, path, (hipStream_t) stream)));
}

int asg_backward(asg_ctx *ctx, const asg_problem *p, const void *state, size_t state_bytes,
                 const void *grad_full, const void *grad_aligned, void *scratch, size_t scratch_bytes,
                 void *grad_transition, void *grad_inputs, int flags, void *stream) {
    (void) ctx; (void) flags;
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (!state || !grad_full || !grad_aligned || !scratch || !grad_transition || !grad_inputs) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    return ASG_DISPATCH(p,
        run_backward<float>(p, state, grad_full, grad_aligned, scratch, scratch_bytes, grad_transition, grad_inputs, 3, (hipStream_t) stream),
        run_backward<double>(p, state, grad_full, grad_aligned, scratch, scratch_bytes, grad_transition, grad_inputs, 3, (hipStream_t) stream));
}

int asg_loss_forward(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes, int reduction,
                     void *loss, void *scores, int flags, void *stream) {
    if (reduction < 0 || reduction > 2 || !loss || !scores) return ASG_ERR_INVALID;
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (!state) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    const size_t e = p->dtype == ASG_DTYPE_F64 ? 8 : 4;
    char *sc = (char *) scores;
    void *full = sc, *ali = sc + (size_t) p->B * e;
    flags &= ~ASG_FLAG_ALPHA_SCORES;
    const bool in_kernel = small_full(p->N) && small_aligned(p->S);
    rc = ASG_DISPATCH(p,
        run_forward<float>(ctx, p, state, full, ali, 15, true, flags, (hipStream_t) stream, loss, reduction),
        run_forward<double>(ctx, p, state, full, ali, 15, true, flags, (hipStream_t) stream, loss, reduction));
    if (rc || in_kernel) return rc;
    return hip_status(ASG_DISPATCH(p,
        launch_loss_reduce<float>(full, ali, (int) p->B, reduction, loss, (hipStream_t) stream),
        launch_loss_reduce<double>(full, ali, (int) p->B, reduction, loss, (hipStream_t) stream)));
}

int asg_loss_forward_only(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes, int reduction,
                          void *loss, void *scores, size_t scores_bytes, int flags, void *stream) {
    if (reduction < 0 || reduction > 2 || !loss || !scores) return ASG_ERR_INVALID;
    int rc = check_problem(p, true);
    if (rc) return rc;
    const size_t e = p->dtype == ASG_DTYPE_F64 ? 8 : 4;
    if (scores_bytes < asg_loss_forward_only_scores_bytes(p)) return ASG_ERR_WORKSPACE;
    const bool in_kernel = small_full(p->N) && small_aligned(p->S);
    if (!in_kernel) {
        if (!state) return ASG_ERR_INVALID;
        if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    } else {
        state = nullptr;
    }
    char *sc = (char *) scores;
    void *full = sc, *ali = sc + (size_t) p->B * e;
    unsigned *ticket = (unsigned *) (sc + align_up(2 * (size_t) p->B * e));
    flags &= ~ASG_FLAG_ALPHA_SCORES;
    rc = ASG_DISPATCH(p,
        run_forward<float>(ctx, p, state, full, ali, kFullBeta | kAlignedBeta, false, flags, (hipStream_t) stream, loss, reduction, ticket),
        run_forward<double>(ctx, p, state, full, ali, kFullBeta | kAlignedBeta, false, flags, (hipStream_t) stream, loss, reduction, ticket));
    if (rc || in_kernel) return rc;
    return hip_status(ASG_DISPATCH(p,
        launch_loss_reduce<float>(full, ali, (int) p->B, reduction, loss, (hipStream_t) stream),
        launch_loss_reduce<double>(full, ali, (int) p->B, reduction, loss, (hipStream_t) stream)));
}

int asg_loss_backward(asg_ctx *ctx, const asg_problem *p, const void *state, size_t state_bytes, int reduction,
                      const void *grad_loss, void *scratch, size_t scratch_bytes, void *grad_transition,
                      void *grad_inputs, int flags, void *stream) {
    (void) ctx; (void) flags;
    int rc = check_problem(p, true);
    if (rc) return rc;
    if (reduction < 0 || reduction > 2) return ASG_ERR_INVALID;
    if (!state || !grad_loss || !scratch || !grad_transition || !grad_inputs) return ASG_ERR_INVALID;
    if (state_bytes < asg_state_bytes(p)) return ASG_ERR_WORKSPACE;
    const int gstride = reduction == 0 ? 1 : 0;
    const double gscale = reduction == 2 ? 1.0 / (double) p->B : 1.0;
    return ASG_DISPATCH(p,
        run_backward<float>(p, state, grad_loss, nullptr, scratch, scratch_bytes, grad_transition, grad_inputs, 3, (hipStream_t) stream, gstride, gscale, 1),
        run_backward<double>(p, state, grad_loss, nullptr, scratch, scratch_bytes, grad_transition, grad_inputs, 3, (hipStream_t) stream, gstride, gscale, 1));
}

/* ---- fused training step (asg_fused.hip) ------------------------------------------------------------------ */

namespace {
struct FusedLayout { size_t tiles, flags, dump, ticket2, p2, edges, ascore, fscore, xstate, aoff, rows, in32, total; };
FusedLayout fused_layout(const asg_problem *p) {
    FusedLayout L{};
    size_t off = 0;
    L.tiles = off; off = align_up(off + (size_t) p->B * 2 * p->N * p->N * 4);
    L.flags = off; off = align_up(off + (size_t) p->B * 4);
    L.dump = off; off = align_up(off + (size_t) p->B * 3 * 4);
    L.ticket2 = off; off = align_up(off + 256);
    L.p2 = off; off = align_up(off + (size_t) p->B * 2 * (p->T + 8) * (p->S < 1 ? 1 : p->S) * 4);
    L.edges = off; off = align_up(off + (size_t) p->B * 2 * 3 * 128 * 8);
    L.ascore = off; off = align_up(off + (size_t) p->B * 8);
    L.fscore = off; off = align_up(off + (size_t) p->B * 8);
    L.xstate = off; off = align_up(off + (size_t) p->B * 2 * ((p->T + 7) / 8 + 2) * 2048);
    L.aoff = off; off = align_up(off + (size_t) p->B * 2 * ((p->T + 15) / 16 + 1) * 2 * 8);
    if (p->inputs_dtype == ASG_DTYPE_BF16) {
        L.rows = off; off = align_up(off + (size_t) p->T * p->B * p->N * 4);
        L.in32 = off; off = align_up(off + (size_t) p->T * p->B * p->N * 4);
    }
    L.total = off;
    return L;
}
FusedArgs fused_args(const asg_problem *p, void *scratch, int reduction) {
    const FusedLayout L = fused_layout(p);
    char *base = (char *) scratch;
    FusedArgs F{};
    F.tiles = base + L.tiles;
    F.flags = (int *) (base + L.flags);
    F.dump = base + L.dump;
    F.ticket2 = (unsigned *) (base + L.ticket2);
    F.p2 = base + L.p2;
    F.edges = base + L.edges;
    F.ascore = base + L.ascore;
    F.fscore = base + L.fscore;
    F.xstate = base + L.xstate;
    F.aoff = base + L.aoff;
    if (p->inputs_dtype == ASG_DTYPE_BF16) { F.rows = base + L.rows; F.in32 = base + L.in32; }
    F.reduction = reduction;
    F.gscale = reduction == 2 ? (float) (1.0 / (double) p->B) : 1.0f;
    return F;
}
}  // namespace

int asg_loss_fused_supported(const asg_problem *p) {
    if (check_problem(p, true, true) != ASG_OK) return 0;
    if (p->dtype != ASG_DTYPE_F32 || p->N >= 64 || p->S > 64) return 0;
    const double fr = (double) (p->T - 1) * (double) p->inputs_strides[0] * 4.0, ln = 63.0 * (double) p->inputs_strides[2] * 4.0;
    if (p->inputs_strides[0] < 0 || p->inputs_strides[2] < 0 || fr >= 4294967296.0 || ln >= 2147483648.0) return 0;
    if ((double) p->T * (double) p->B * (double) p->N * 4.0 >= 4294967296.0) return 0;
    if (p->T > 4000) return 0;              // per-block offset tables of the aligned finishers (asg_fused.hip: kMaxBlk)
    return 1;
}

size_t asg_loss_fused_scratch_bytes(const asg_problem *p) {
    if (!p || p->T < 1 || p->B < 1 || p->N < 1) return 0;
    return fused_layout(p).total;
}

size_t asg_loss_fused_sync_bytes(const asg_problem *p) {
    if (!p || p->B < 1) return 0;
    return align_up(256 + (size_t) p->B * 64);
}

int asg_loss_fused_forward(const asg_problem *p, void *state, size_t state_bytes, int reduction, void *loss, void *scores,
                           void *scratch, size_t scratch_bytes, void *grad_inputs, void *sync, int flags, void *stream) {
    (void) flags;
    if (reduction < 0 || reduction > 2 || !loss || !scores || !scratch || !grad_inputs || !sync || !state) return ASG_ERR_INVALID;
    if (!asg_loss_fused_supported(p)) return check_problem(p, true, true) != ASG_OK ? check_problem(p, true, true) : ASG_ERR_UNSUPPORTED;
    if (state_bytes < asg_state_bytes(p) || scratch_bytes < asg_loss_fused_scratch_bytes(p)) return ASG_ERR_WORKSPACE;
    FusedArgs F = fused_args(p, scratch, reduction);
    F.loss = loss;
    F.scores = scores;
    F.grad_inputs = grad_inputs;
    if (!F.rows) F.rows = grad_inputs;
    F.sync = (unsigned *) sync;
    return hip_status(launch_fused_forward(to_problem(p), to_state(p, state), F, (hipStream_t) stream));
}

int asg_loss_fused_backward(const asg_problem *p, void *state, size_t state_bytes, int reduction, const void *grad_loss,
                            void *scratch, size_t scratch_bytes, void *grad_inputs, void *grad_transition, int flags,
                            void *stream) {
    (void) flags;
    if (reduction < 0 || reduction > 2 || !grad_loss || !scratch || !grad_inputs || !grad_transition || !state) return ASG_ERR_INVALID;
    if (!asg_loss_fused_supported(p)) return check_problem(p, true, true) != ASG_OK ? check_problem(p, true, true) : ASG_ERR_UNSUPPORTED;
    if (state_bytes < asg_state_bytes(p) || scratch_bytes < asg_loss_fused_scratch_bytes(p)) return ASG_ERR_WORKSPACE;
    FusedArgs F = fused_args(p, scratch, reduction);
    F.grad_inputs = grad_inputs;
    if (!F.rows) F.rows = grad_inputs;
    F.grad_loss = grad_loss;
    F.grad_transition = grad_transition;
    return hip_status(launch_fused_backward(to_problem(p), to_state(p, state), F, (hipStream_t) stream));
}

}  // extern "C"
