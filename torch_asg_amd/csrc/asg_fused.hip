// torch_asg_amd/csrc/asg_fused.hip -- the fused training step of the small-alphabet path (fp32, N < 64, S <= 64):
// ONE launch runs the four recursions of every utterance AND assembles its gradients as the frames become ready.
//
// Replaces, for the ASGLoss training route, fast_asg_gpu_forward + fast_asg_gpu_backward
// (/root/reference/torch_asg/native/streamlined_fast_gpu.cpp:104-297) and everything under them
// (fully_connected_lattice.cpp:9-105, force_aligned_lattice.cpp:15-356, force_aligned_lattice_kernel.cu): the reference's
// split "both recursions in forward, no recursion in backward" is kept and taken one step further -- the
// non-recursive assembly happens inside the forward launch too, and backward only scales by the upstream gradient.
//
// Three workgroups per utterance, each alone on a compute unit (grid = 48 * ceil(B / 16), 512 threads; the three of
// an utterance have block indices that are equal mod 8, which puts them behind the same L2 in practice -- speed only):
//   * the ALIGNED workgroup owns the force-aligned lattice: both chains, their crossing, the aligned posteriors and the
//     edge posteriors.  It depends on nothing and never waits for another workgroup.
//   * the FULL-ALPHA and FULL-BETA workgroups each own one direction of the fully-connected lattice.  One compute unit
//     per direction because the recursion's LDS broadcast (10 ds_read_b128 per step) takes a third of a compute unit's
//     LDS bandwidth: with both directions behind one LDS the step was 317 cycles instead of 240 (measured, round 2).
// The alpha chain walks frames 0 -> len-1 and the beta chain len-1 -> 0; they cross at mid = len/2.  Before the
// crossing each chain stores its state for the other side (frames < mid from alpha, >= mid from beta: half of what the
// stand-alone kernels store; write-through 16-byte stores, four frames per lane and store, behind ONE progress word);
// after it, the side that reaches a frame SECOND holds everything the frame's gradient needs:
//   posterior_t = softmax(alpha_t + beta_t)                      -> grad_inputs row (minus the scattered aligned posterior)
//   alpha side:  xi_t(i,j)    = posterior_t[i]   / s_i  * E[i][j] * v_{t-1}[j]     s = E v_{t-1}  (this step's row sums)
//   beta side:   xi_{t+1}(i,j) = posterior_t[j] / s'_j * F[j][i] * y_{t+1}[i]      s' = F y_{t+1}
// i.e. the recursion's OWN row sums and broadcast vector -- the stand-alone assembly kernel's second mat-vec is gone, and
// the sum over frames of the outer products (posterior / s) (x) v runs on the matrix cores (asg_outer.h).
// Wave roles of a full workgroup (waves with equal index % 4 share a SIMD; the recursion wavefront has SIMD 0 to itself):
//   0  recursion wavefront (critical path only; as fwd_duo_kernel)        1  producer (emission factors)
//   2, 3, 6 (, 7)  consumers: first half = log-domain state -> HBM (consumer 0); second half = full-lattice posterior ->
//         grad_inputs row, xi accumulation (MFMA, double accumulators), taking the groups of 8 frames round-robin
// of the aligned workgroup:  0/1 chains alpha/beta   2,6,5 / 3,7,4 finishers (aligned posterior -> HBM, edge posteriors)
// The backward launch finishes the grad_inputs rows: minus the aligned posteriors scattered to labels, times the upstream
// gradient.
// Everything is bit-deterministic: no float atomics, fixed accumulation orders.
// An utterance whose row sums leave the safe range (or shorter than kMinFused frames, or any bounded wait that runs
// out) is FLAGGED: its scores are recomputed here with exact log-sum-exps, its gradients by the exact stand-alone code
// in the backward launch (fused_bwd_kernel), so results are true log-sum-exps for any input.
#include "asg_assemble.h"
#include "asg_outer.h"

namespace asg {
namespace {

constexpr int kAR = 64;         // aligned chain -> finisher ring of aligned states (frames)
constexpr int kGS = 8;          // frames per poll of the consumers / finishers
constexpr int kMinFused = 4;
constexpr int kAF = 3;          // aligned finisher wavefronts per side (round-robin over 8-index groups)
// consumer wavefronts per full workgroup (round-robin over 8-index groups of the second half): what the LDS holds
constexpr int kMaxNC = 4;
template <int NP> struct Consumers {          // each has a [16 NT][16 NT + 1] double tile beside the 64 KB of rings
    static constexpr int fit = NP <= 48 ? 4 : 2;
    static constexpr int n = 3 < fit ? 3 : fit;
};
// Developer variants (tests/test_hip_variants.py builds them through build.py --define; the shipped library has neither):
//   ASG_X_SPREAD_XCD     the three workgroups of an utterance on three DIFFERENT XCDs: every cross-workgroup hand-off
//                        then crosses L2s, which the default placement (all behind one L2, speed only) never exercises
//   ASG_X_TEST_DELAY     utterance 1's aligned workgroup and utterance 2's full-alpha workgroup start late, past every
//                        bounded wait of their partners (caps divided by 2^ASG_X_CAPSHIFT so the test stays short)
#ifndef ASG_X_CAPSHIFT
#define ASG_X_CAPSHIFT 0
#endif
constexpr int kCapLds = kSpinCap >> ASG_X_CAPSHIFT;               // waits on a word of this workgroup's LDS
constexpr int kCapGlobal = (kSpinCap >> 4) >> ASG_X_CAPSHIFT;     // waits on another workgroup's progress word
constexpr int kCapVerdictEarly = (1 << 18) >> ASG_X_CAPSHIFT;     // the early look at the aligned workgroup's verdict
constexpr int kCapVerdict = (1 << 24) >> ASG_X_CAPSHIFT;          // the epilogue's wait for it
constexpr unsigned kClosedWithoutAligned = 3u;                    // UttSync::adone: the full workgroups gave up waiting
constexpr int kFusedThreads = 512;   // 8 wavefronts: two per SIMD, 256 VGPRs each (the consumers keep 72 of double accumulators)
constexpr unsigned kSc1 = 16;   // buffer load/store aux bit: agent scope (served by / written through to L2)

constexpr int kFR = 64;                                   // ring depth of the full workgroup (frames): two consumers that
                                                          // alternate need more slack than the 32 of the stand-alone kernel
struct FusedSide {                                        // one direction (alpha / beta) of the FULL workgroup
    static constexpr int kR = kFR;
    __attribute__((aligned(16))) float p[kFR][64];     // step n's broadcast vector v_n (slot n & 63)
    float s[kFR][64];                                    // row sums s_n, main -> consumer (self-describing: NaN sentinel)
    float e[kFR][64];                                    // emission factors, producer -> main
    float a[kFR][64];                                    // their log2 (alpha side), producer -> consumer
    float x[64];                                         // row / column maxima of the transition matrix
    double zsum;
    int e_prod, csum, main_done, prod_done, kill;
    // kNC consumer wavefronts: wave 0 takes the whole first half; in the second half wave k takes the 8-index groups
    // k, k + kNC, ...  cd[k] = (last index of wave k's latest group) + 8 (kNC - 1): everything up to the minimum over k has
    // been taken out of the rings (waves 1.. count as "infinitely far" during the first half).
    int cd[kMaxNC];
    int st_done;      // consumer 0: state rows of indices [0, st_done) are in HBM/L2 and visible
    __device__ __forceinline__ int consumed() {
        int v = __hip_atomic_load(&cd[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 1; k < kMaxNC; ++k) v = min(v, __hip_atomic_load(&cd[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        return v;
    }
    __device__ __forceinline__ float *pslot(int n) { return p[n & (kFR - 1)]; }
    __device__ __forceinline__ bool stop() { return __hip_atomic_load(&kill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; }
};

constexpr int kMaxBlk = 128;                              // blocks of 16 indices per half whose offsets the finisher keeps in LDS
struct AliSide {                                          // one direction of the ALIGNED workgroup
    float ar[kAR][64];                                   // aligned states, chain -> finisher (slot (index - 1) & (kAR - 1))
    double cb[4][2];                                     // per 16-index block (slot j & 3): offset at block entry, per-index step
    double ob[kMaxBlk][2];                               // finisher: the OTHER side's first-half block offsets (from HBM, once);
                                                         // entry j + 1 = block j, entry 0 = {0, 0} for index 0
    int kill;
    int ast_done;     // chain: states of indices [0, ast_done) are in HBM/L2 and visible
    int ar_done;      // chain: states of indices [0, ar_done) have been written to `ar`
    int fd[kAF];      // finisher k: (count of indices through its latest group) + 8 * (kAF - 1); ring slots are free up to
                      // the minimum over k
    __device__ __forceinline__ int finished() {
        return min(min(__hip_atomic_load(&fd[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                       __hip_atomic_load(&fd[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)),
                   __hip_atomic_load(&fd[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    __device__ __forceinline__ bool stop() { return __hip_atomic_load(&kill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; }
};

// Cross-workgroup words of one utterance (FusedArgs::sync, after the 64-word ticket block).  Zero on entry, zero on exit.
struct UttSync {
    unsigned adone;         // aligned workgroup: 1 = finished (edges + score written), 2 = finished but gave up
    unsigned st_done[2];    // full workgroup alpha / beta: the states of its indices [0, st_done) are visible (xstate)
    unsigned kill;          // any of the three workgroups gave up on the fused path for this utterance
    unsigned arrive;        // full workgroups that have finished (the second one closes the utterance)
    unsigned pad[11];
};
static_assert(sizeof(UttSync) == 64, "asg_loss_fused_sync_bytes");

// First-half states of a full chain for the other side: blocks of 8 indices, [block][quad][lane][4 indices] floats, so a
// lane stores / loads four consecutive indices of its label with ONE 16-byte write-through access.  Index m of a side
// with first-half length h sits at position m + ((-h) & 7): the LAST block of the half is full and aligned, and the
// other side -- which walks these indices downwards, 8 per group, starting from the last -- reads whole blocks.
// Where the two chains of an utterance cross.  Near the middle, placed so that the LAST 8-frame group of both second
// halves is short: what is left to do when the recursions end is one group's latency, and a group of <= 4 frames takes
// half of it.  (All three workgroups and the backward launch must agree on it.)
__host__ __device__ __forceinline__ int crossing(int len) {
    int mid = len / 2;
    if (len < 64) return mid;
    const int base = (mid >> 3) << 3;
    int best = mid, cost = 99;
    for (int r = 1; r <= 5; ++r) {
        const int m = base + r, ra = (len - m) & 7, c = max(r, ra == 0 ? 8 : ra);
        if (c < cost) { cost = c; best = m; }
    }
    return best;
}

constexpr int kXBlockBytes = 2 * 64 * 16;
__host__ __device__ __forceinline__ int xstate_blocks(int T) { return (T + 7) / 8 + 2; }

template <int NP>
struct TileLds {
    double sx[Consumers<NP>::n][16 * ((NP + 15) / 16)][16 * ((NP + 15) / 16) + 1];   // (padded to whole MFMA tiles: the dump needs no bounds checks)   // xi sums of consumer 0 .. kNC-1 (double: asg_outer.h): alpha side [to i][from j], beta side [from j][to i]
                                        // (before the E / F factor)
};
template <int NP>
struct EdgeLds {
    unsigned long long fxT[NP * NP];    // aligned edge posteriors of this side's frames, fixed point, [to][from]
};

template <int NP>
struct FusedShared {
    union U {
        FusedSide g;
        struct H { AliSide A, B; } h;
        EdgeLds<NP> e;       // epilogue only: the rings are dead by then
    } u;
    TileLds<NP> t;       // beside the rings, not over them: each consumer adds its accumulators as its LAST act, so they are
                         // not live across the roles (as values handed to the epilogue they were spilled inside the loop)
    double score_full, score_ali;
    int adone, last;
    int tr_ready;                 // wavefronts that have written their share of `trl`
    float xs[64];
    float trl[NP * NP];           // full workgroups: the transition matrix, compact [N][N] (filled by the idle wavefronts while
                                  // the producer's first emission loads are in flight; read by the recursion wavefront, the
                                  // producer and the epilogue instead of three rounds of strided global loads)
    // one workgroup per compute unit (160 KB of LDS): see the header
    char pad[(sizeof(U) + sizeof(TileLds<NP>) + 4 * NP * NP + 512 < 84 * 1024) ? 84 * 1024 - sizeof(U) - sizeof(TileLds<NP>) - 4 * NP * NP - 512 : 16];
};

// ---- kernel parameters ------------------------------------------------------------------------------------------
// The three parameter blocks are ~80 SGPRs.  Left as ordinary by-value kernel parameters they are loaded at kernel
// entry and kept live through every role (they are needed again in the epilogue), and the recursion loops then run on
// spilled scalars (v_readlane per use).  Instead every role re-reads what it needs from the kernarg segment through a
// pointer the compiler cannot see through, so the values live only inside that role.
struct FusedParams {
    Problem P;
    State W;
    FusedArgs F;
};
typedef const FusedParams __attribute__((address_space(4))) *KParams;

__device__ __forceinline__ KParams kernarg_params() {
    KParams p = (KParams) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ Problem ld_problem(KParams k) {
    Problem P;
    P.inputs = k->P.inputs; P.is0 = k->P.is0; P.is1 = k->P.is1; P.is2 = k->P.is2;
    P.transition = k->P.transition; P.ts0 = k->P.ts0; P.ts1 = k->P.ts1;
    P.targets = k->P.targets; P.gs0 = k->P.gs0; P.gs1 = k->P.gs1;
    P.in_len = k->P.in_len; P.tg_len = k->P.tg_len;
    P.T = k->P.T; P.B = k->P.B; P.N = k->P.N; P.S = k->P.S; P.in_bf16 = k->P.in_bf16;
    return P;
}
__device__ __forceinline__ State ld_state(KParams k) {
    State W;
    W.ah = k->W.ah; W.bh = k->W.bh; W.ab = k->W.ab; W.bb = k->W.bb;
    W.ehat = k->W.ehat; W.fhat = k->W.fhat; W.rmax = k->W.rmax; W.cmax = k->W.cmax;
    W.asu = k->W.asu; W.asi = k->W.asi; W.dbg = k->W.dbg; W.ticket = k->W.ticket; W.work = k->W.work;
    W.npad = k->W.npad;
    return W;
}
__device__ __forceinline__ FusedArgs ld_fargs(KParams k) {
    FusedArgs F;
    F.loss = k->F.loss; F.scores = k->F.scores; F.grad_inputs = k->F.grad_inputs; F.tiles = k->F.tiles;
    F.flags = k->F.flags; F.dump = k->F.dump; F.p2 = k->F.p2; F.edges = k->F.edges; F.ascore = k->F.ascore;
    F.aoff = k->F.aoff; F.sync = k->F.sync; F.xstate = k->F.xstate; F.fscore = k->F.fscore; F.rows = k->F.rows; F.in32 = k->F.in32; F.ticket2 = k->F.ticket2; F.grad_loss = k->F.grad_loss;
    F.grad_transition = k->F.grad_transition; F.reduction = k->F.reduction; F.gscale = k->F.gscale;
    return F;
}

// developer probes (-DASG_PROBE): per role of utterance 0, total cycles and cycles spent in each kind of wait
#ifdef ASG_PROBE
#define PRB_DECL const long long prb_t0 = clock64(); long long prb_w[4] = {0, 0, 0, 0};
#define PRB_WAIT(i, expr) { const long long prb_a = clock64(); expr; prb_w[i] += clock64() - prb_a; }
#define PRB_END(dbgp, role) if (b == 0 && (threadIdx.x & 63) == 0) { long long *d = (long long *) (dbgp) + (role) * 5; \
    d[0] = clock64() - prb_t0; d[1] = prb_w[0]; d[2] = prb_w[1]; d[3] = prb_w[2]; d[4] = prb_w[3]; }
#else
#define PRB_DECL
#define PRB_WAIT(i, expr) { expr; }
#define PRB_END(dbgp, role)
#endif

#ifdef ASG_PROBE
__device__ unsigned g_abort_code[4];      // developer builds: site number of the last abort, and how many there were
#endif
__device__ __forceinline__ void note_abort(int site) {
#ifdef ASG_PROBE
    if ((threadIdx.x & 63) == 0) { g_abort_code[0] = (unsigned) site; atomicAdd(&g_abort_code[1], 1u); g_abort_code[2] = blockIdx.x; }
#endif
}

// How a role gives up / learns that somebody else has.  Inside a workgroup the word is in LDS (polled by every bounded
// wait); across the three workgroups of an utterance it is UttSync::kill, polled by every wait on a global word.
struct FullCtl {
    FusedSide *L;
    UttSync *us;
    __device__ __forceinline__ bool stop() const { return L->stop(); }
    __device__ __forceinline__ void abort(int site) const {
        lds_store_rlx(&L->kill, 1);
        __hip_atomic_store(&us->kill, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        note_abort(site);
    }
    __device__ __forceinline__ bool killed_elsewhere() const {
        return __hip_atomic_load(&us->kill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
};
struct AliCtl {
    AliSide *L, *O;
    UttSync *us;
    __device__ __forceinline__ bool stop() const { return L->stop(); }
    __device__ __forceinline__ void abort(int site) const {
        lds_store_rlx(&L->kill, 1);
        lds_store_rlx(&O->kill, 1);
        __hip_atomic_store(&us->kill, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        note_abort(site);
    }
};

// bounded wait until *p >= need; false on abort / time-out (then everything is aborted)
template <class Ctl>
__device__ __forceinline__ bool wait_ge(int *p, int need, const Ctl &c) {
    int spins = 0;
    while (lds_load_rlx(p) < need) {
        if (c.stop()) return false;
        if (++spins > kCapLds) { c.abort(1); return false; }
        __builtin_amdgcn_s_sleep(6);
    }
    asm volatile("" ::: "memory");
    return true;
}

// bounded wait until L.finished() >= need
template <class Ctl>
__device__ __forceinline__ bool wait_finished(int need, const Ctl &c) {
    int spins = 0;
    while (c.L->finished() < need) {
        if (c.stop()) return false;
        if (++spins > kCapLds) { c.abort(2); return false; }
        __builtin_amdgcn_s_sleep(6);
    }
    asm volatile("" ::: "memory");
    return true;
}

// v_rcp_f32 is good to 1 ulp but not unbiased: sums of thousands of posteriors that should cancel exactly (tiny alphabets:
// the full-lattice and the aligned edge posteriors are the same numbers) see the bias.  One Newton step removes it.
__device__ __forceinline__ float rcp_nr(float x) {
    const float r = Num<float>::rcp(x);
    const float e = fmaf(-x, r, 1.0f);              // NaN for x = 0, inf (r = inf, 0): keep r then
    return (e == e) ? fmaf(e, r, r) : r;
}

// a / b correctly rounded (b finite, nonzero, well inside the fp32 range): v_rcp_f32 and one residual step ON THE QUOTIENT.
// a * rcp(b) alone is biased where it matters most: for a == b (the posterior of a one-label alphabet) it gives 1 or
// 1 - 2^-24, never 1 + anything, and thousands of such terms no longer cancel against the aligned lattice's.
__device__ __forceinline__ float div_nr(float a, float b) {
    const float r = Num<float>::rcp(b);
    const float q = a * r;
    return fmaf(fmaf(-q, b, a), r, q);
}

// float -> bfloat16, round to nearest even (gradients are finite)
__device__ __forceinline__ unsigned short float_to_bf16_bits(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short) (u >> 16);
}
// bfloat16 emissions: the exact stand-alone code (flagged utterances) reads fp32 -- utterance b's emissions, widened, in
// FusedArgs::in32 ([B][T][N]) and the Problem that points the kernels there
__device__ __forceinline__ Problem problem_fp32_copy(const Problem &P, const FusedArgs &F) {
    Problem Q = P;
    Q.inputs = F.in32;
    Q.is0 = P.N; Q.is1 = (int64_t) P.T * P.N; Q.is2 = 1;
    Q.in_bf16 = 0;
    return Q;
}
__device__ __forceinline__ void widen_utterance(const Problem &P, const FusedArgs &F, int b, int nthreads) {
    const unsigned short *src = (const unsigned short *) P.inputs + (int64_t) b * P.is1;
    float *dst = (float *) F.in32 + (int64_t) b * P.T * P.N;
    for (int k = threadIdx.x; k < P.T * P.N; k += nthreads) {
        const int t = k / P.N, i = k - t * P.N;
        dst[k] = bf16_bits_to_float(src[(int64_t) t * P.is0 + (int64_t) i * P.is2]);
    }
}

__device__ __forceinline__ float buf_load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, kSc1));
}
__device__ __forceinline__ void buf_store_sc1(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, kSc1);
}
// The launches' bulk outputs (gradient rows, aligned posteriors) are written THROUGH the L2 (sc0 sc1) as they are produced instead of
// being left dirty for the write-back at the end of the kernel: -0.5 us on the forward launch, -0.2 us on the backward launch at
// cfg 3 (63.9 -> 63.2 us per step; A/B with tools/fused_ab.sh; nt: no gain, backward +0.6 us).  -DASG_X_OUT_AUX=0: plain stores.
#ifndef ASG_X_OUT_AUX
#define ASG_X_OUT_AUX 17
#endif
__device__ __forceinline__ void buf_store_out(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, ASG_X_OUT_AUX);
}
// global progress word of another workgroup: bounded relaxed agent-scope poll (one lane's value, uniform); the
// utterance's kill word (same cache line) is polled with it
__device__ __forceinline__ bool wait_global_ge(unsigned *p, unsigned need, unsigned &seen, const FullCtl &c) {
    int spins = 0;
    while (seen < need) {
        seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (seen >= need) break;
        if (c.stop()) return false;
        if (c.killed_elsewhere()) { c.abort(12); return false; }
        if (++spins > kCapGlobal) { c.abort(3); return false; }
        __builtin_amdgcn_s_sleep(16);
    }
    return true;
}

// ------------------------------------------------------------------ recursion wavefront
template <int NP, bool BETA>
__device__ __forceinline__ void fused_main(const Problem &P, int b, FusedSide &L, UttSync *us, int len, void *dbg,
                                           const float *tr_lds, int *tr_ready, int tr_need) {
    const FullCtl ctl{&L, us};
    typedef float R;
    PRB_DECL
#ifdef ASG_PROBE
    const long long prb_wall0 = (long long) wall_clock64();
#endif
    const int lane = threadIdx.x & 63;
    const int N = P.N;
    const bool act = lane < N;
    const int lc = act ? lane : 0;
    {
        int spins = 0;
        while (__hip_atomic_load(tr_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < tr_need) {
            if (L.stop()) return;
            if (++spins > kCapLds) { ctl.abort(14); return; }
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    }
    const R *tline = tr_lds + lc * (BETA ? 1 : N);
    V2<R> e2[NP / 2];
    R X;
    load_norm_row<R, NP>(tline, BETA ? N : 1, N, act, e2, X);
    if (lane == N) {
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) e2[j] = V2<R>{1, 1};
    }
    const int nst = len - 1;                   // mat-vecs n = 0 .. len-2
    R s_prev = act ? Num<R>::exp2(-X) : R(0);
    int csum = 0;
    bool next_ready = false;
    R e_next_first = 0;
    for (int n0 = 0; n0 < nst; n0 += kPF) {
        const int nsteps = min(kPF, nst - n0);
        const int need = min(n0 + kPF, len);
        int spins = 0;
        R e_first = e_next_first;
        PRB_WAIT((n0 < len / 2 ? 0 : 1), while (!next_ready) {
            const int ep = lds_load_rlx(&L.e_prod);
            const int kl = lds_load_rlx(&L.kill);
            e_first = lds_ldf(&L.e[n0 & (kFR - 16)][lane]);
            asm volatile("" ::: "memory");
            if (kl) return;
            if (ep >= need) break;
            if (++spins > kCapLds) { ctl.abort(4); return; }
            __builtin_amdgcn_s_sleep(1);
        })
        const int need_next = min(n0 + 2 * kPF, len);
        if (nsteps == kPF)
            duo_main_block<NP, false>(L, n0, kPF, e2, N, lane, e_first, s_prev, csum, need_next, next_ready, e_next_first);
        else
            duo_main_block<NP, true>(L, n0, nsteps, e2, N, lane, e_first, s_prev, csum, need_next, next_ready, e_next_first);
    }
    lds_stf(&L.s[(nst - 1) & (kFR - 1)][lane], s_prev);
    lds_store_rlx(&L.csum, csum);
    lds_store_rel(&L.main_done, 1);
#ifdef ASG_PROBE_TAIL
    if (b == 0 && !BETA && lane == 0) ((long long *) dbg)[40] = clock64();
#endif
#ifdef ASG_PROBE
    prb_w[2] = (long long) wall_clock64() - prb_wall0;        // 100 MHz ticks: the shader clock this launch really ran at
#endif
    PRB_END(dbg, BETA ? 1 : 0)
}

// ------------------------------------------------------------------ consumer / full-lattice assembler
// Index convention of a side: index m = 0 .. len-1 is frame f_m = m (alpha) or len-1-m (beta); s_{m-1} are the row sums
// that lead to index m, v_{m-1} (= p slot m-1) the vector that produced them.  Indices < h are the side's first half.
template <int NP, bool BETA>
__device__ __forceinline__ void fused_consumer(const Problem &P, const State &W, const FusedArgs &F, int b, FusedSide &L,
                                               UttSync *us, int len, int h,
                                               double (&sx)[16 * ((NP + 15) / 16)][16 * ((NP + 15) / 16) + 1], double &score_out2, const int cw) {
    typedef float R;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    constexpr int NT = (NP + 15) / 16;
    constexpr int kNC = Consumers<NP>::n;
    const FullCtl ctl{&L, us};
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const R NINF = Num<R>::ninf();
    const bool act = lane < N;
    const unsigned long long actmask = __ballot(act);
    const int lc = act ? lane : 0;
    // exchanged first-half states (see kXBlockBytes): own region written in the first half, the other side's read in the second
    const int xb = xstate_blocks(T);
    char *xbase = (char *) F.xstate + (int64_t) b * 2 * xb * kXBlockBytes;
    __amdgpu_buffer_rsrc_t rs = make_rsrc(xbase + (int64_t) (BETA ? 1 : 0) * xb * kXBlockBytes, (unsigned) xb * kXBlockBytes);
    __amdgpu_buffer_rsrc_t ro = make_rsrc(xbase + (int64_t) (BETA ? 0 : 1) * xb * kXBlockBytes, (unsigned) xb * kXBlockBytes);
    const unsigned voff = act ? (unsigned) lane * 16u : kOobOffset;
    const unsigned vld = (unsigned) lc * 16u;
    const int phi = (-h) & 7, phiO = (-(len - h)) & 7;
    const R gscale = F.gscale;
    // rows of the full-lattice posterior go straight to grad_inputs; the backward launch subtracts the aligned posteriors
    __amdgpu_buffer_rsrc_t rs_g = make_rsrc((R *) F.rows + (int64_t) b * N,
                                            (unsigned) ((int64_t) (T - 1) * P.B * N + N) * (unsigned) sizeof(R));
    const unsigned voffg = act ? (unsigned) lane * (unsigned) sizeof(R) : kOobOffset;
    const unsigned grow_bytes = (unsigned) P.B * N * sizeof(R);
    auto frame = [&](int m) { return BETA ? len - 1 - m : m; };
    // xi sums in DOUBLE accumulators (asg_outer.h): tile (r, c), register q, lane l = element (16 r + (l >> 4) + 4 q, 16 c + (l & 15))
    V4d acc[NT * NT];
#pragma unroll
    for (int q = 0; q < NT * NT; ++q) acc[q] = V4d{0, 0, 0, 0};
    score_out2 = -1e300;
#ifdef ASG_PROBE
    long long prb_seg[5] = {0, 0, 0, 0, 0};
#endif
    PRB_DECL
    if (!wait_ge(&L.e_prod, 1, ctl)) return;                  // X and block 0 of the rings are there
    const R XX = lds_ldf(&L.x[lane]);
    R sv = act ? Num<R>::exp2(-XX) : R(0);
    auto wait_slot = [&](int m) {                             // until main has written s_m (main writes in order)
        float *slot = &L.s[m & (kFR - 1)][lane];
        int spins = 0;
        while (true) {
            const R v = lds_ldf(slot);
            if (__ballot(__float_as_uint(v) != kSentinel) == ~0ull) return true;
            if (L.stop()) return false;
            if (++spins > kCapLds) { ctl.abort(5); return false; }
            __builtin_amdgcn_s_sleep(3);          // ~1 recursion step: every poll is an LDS access the recursion waits behind
        }
    };
    if (cw == 0) {
        // ---- first half (consumer 0 only): log-domain state of indices 0 .. h-1 for the other side, 8 positions per block
        const R v0 = (BETA ? XX : lds_ldf(&L.a[0][lane])) + Num<R>::log2(sv);
        const int last_blk = (h - 1 + phi) >> 3;
        for (int K = 0; K <= last_blk; ++K) {
            const int lo = max(8 * K - phi, 1), hi = min(8 * K + 7 - phi, h - 1);     // ring indices of the block
            R val[kGS];
            if (hi >= lo) {
                PRB_WAIT(0, if (!wait_slot(hi - 1)) return;)
                R sg[kGS], ag[kGS];
#pragma unroll
                for (int q = 0; q < kGS; ++q) {
                    const int m = min(max(8 * K + q - phi, lo), hi);
                    sg[q] = lds_ldf(&L.s[(m - 1) & (kFR - 1)][lane]);
                    ag[q] = BETA ? XX : lds_ldf(&L.a[m & (kFR - 1)][lane]);
                }
                unsigned rlo = 0xffffffffu, rhi = 0;
#pragma unroll
                for (int q = 0; q < kGS; ++q) {
                    const int m = min(max(8 * K + q - phi, lo), hi);
                    lds_stf(&L.s[(m - 1) & (kFR - 1)][lane], __uint_as_float(kSentinel));
                    const unsigned sb = Rng<R>::bits(sg[q]);
                    rlo = min(rlo, sb);
                    rhi = max(rhi, sb);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lds_store_rlx(&L.cd[0], hi);
                if ((__ballot(rlo < Rng<R>::lo || rhi > Rng<R>::hi) & actmask) != 0) { ctl.abort(6); return; }
#pragma unroll
                for (int q = 0; q < kGS; ++q) val[q] = ag[q] + Num<R>::log2(sg[q]);
                sv = sg[kGS - 1];
            } else {
#pragma unroll
                for (int q = 0; q < kGS; ++q) val[q] = v0;
            }
            if (K == 0) {
#pragma unroll
                for (int q = 0; q < kGS; ++q) val[q] = (q == phi) ? v0 : val[q];
            }
            const u4 lo4 = {__float_as_uint(val[0]), __float_as_uint(val[1]), __float_as_uint(val[2]), __float_as_uint(val[3])};
            const u4 hi4 = {__float_as_uint(val[4]), __float_as_uint(val[5]), __float_as_uint(val[6]), __float_as_uint(val[7])};
            __builtin_amdgcn_raw_buffer_store_b128(lo4, rs, voff, (unsigned) K * kXBlockBytes, kSc1);
            __builtin_amdgcn_raw_buffer_store_b128(hi4, rs, voff, (unsigned) K * kXBlockBytes + 1024u, kSc1);
        }
        // from here on the other consumers count: consumer k's first own group starts at index h + 8 k
#pragma unroll
        for (int k = 1; k < kNC; ++k) lds_store_rlx(&L.cd[k], h - 1 + k * kGS);
        // the first half has been written through before the other side (and consumer 1) is told so
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&us->st_done[BETA ? 1 : 0], (unsigned) h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_store_rel(&L.st_done, h);
    } else {
        PRB_WAIT(0, if (!wait_ge(&L.st_done, h, ctl)) return;)
    }
    int n = h + cw * kGS;                 // this wavefront's first group of the second half
    {
        unsigned seen = 0;
        PRB_WAIT(1, if (!wait_global_ge(&us->st_done[BETA ? 0 : 1], (unsigned) (len - h), seen, ctl)) return;)
    }
    // the other side's states of own indices nn .. nn+7 = ITS indices len-1-nn .. len-8-nn = one whole block (see above);
    // write-through stores there, L2-served loads here: no fence on either side
    auto load_other = [&](int nn, R (&dst)[kGS]) {
        const int Kr = max((len - 1 - nn + phiO) >> 3, 0);
        const u4 a = __builtin_amdgcn_raw_buffer_load_b128(ro, vld, (unsigned) Kr * kXBlockBytes, kSc1);
        const u4 c = __builtin_amdgcn_raw_buffer_load_b128(ro, vld, (unsigned) Kr * kXBlockBytes + 1024u, kSc1);
        dst[7] = __uint_as_float(a.x); dst[6] = __uint_as_float(a.y); dst[5] = __uint_as_float(a.z); dst[4] = __uint_as_float(a.w);
        dst[3] = __uint_as_float(c.x); dst[2] = __uint_as_float(c.y); dst[1] = __uint_as_float(c.z); dst[0] = __uint_as_float(c.w);
    };
    // ---- second half: indices h .. len-1; the other side's state of the same frames is prefetched one group ahead
    R oth[kGS];
    load_other(n, oth);
    while (n < len) {
        const int g = min(kGS, len - n);
        // ring data of the group in ONE LDS round trip: main writes s in order, so once the group's LAST row sum is
        // there (no sentinel) everything read with it is valid; otherwise sleep and read again
        R sg[kGS], ag[kGS], pg[kGS];
#ifdef ASG_PROBE
        const long long prb_slot0 = clock64();
#endif
#ifdef ASG_PROBE_TAIL
        long long tl[8]; tl[0] = clock64();
#endif
        // every poll is an LDS access the recursion wavefront's broadcast reads queue behind, and kNC - 1 consumers are
        // waiting at any time: sleep in long steps until the group's FIRST row sum is there, then through most of the
        // seven recursion steps that follow, and only then poll closely
        if (g > 2 && n + g < len) {
            float *first = &L.s[(n - 1) & (kFR - 1)][lane];
            int spins = 0;
            while (__ballot(__float_as_uint(lds_ldf(first)) != kSentinel) != ~0ull) {
                if (L.stop()) return;
                if (++spins > kCapLds) { ctl.abort(13); return; }
                __builtin_amdgcn_s_sleep(10);
            }
            __builtin_amdgcn_s_sleep(2 * (kGS - 2));
        }
        if (!wait_slot(n + g - 2)) return;
#ifdef ASG_PROBE_TAIL
        tl[1] = clock64();
#endif
#pragma unroll
        for (int q = 0; q < kGS; ++q) {
            const int m = n + min(q, g - 1);
            sg[q] = lds_ldf(&L.s[(m - 1) & (kFR - 1)][lane]);
            ag[q] = BETA ? XX : lds_ldf(&L.a[m & (kFR - 1)][lane]);
            pg[q] = lds_ldf(&L.p[(m - 1) & (kFR - 1)][lane]);
        }
#ifdef ASG_PROBE
        prb_w[2] += clock64() - prb_slot0;
        const long long seg_a = clock64();
#endif
        unsigned lo = 0xffffffffu, hi = 0;
#pragma unroll
        for (int q = 0; q < kGS; ++q) {
            const int m = n + min(q, g - 1);
            lds_stf(&L.s[(m - 1) & (kFR - 1)][lane], __uint_as_float(kSentinel));
            const unsigned sb = Rng<R>::bits(sg[q]);
            lo = min(lo, sb);
            hi = max(hi, sb);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the values are in registers before the producer may refill
        lds_store_rlx(&L.cd[cw], n + g - 1 + (kNC - 1) * kGS);
        if ((__ballot(lo < Rng<R>::lo || hi > Rng<R>::hi) & actmask) != 0) { ctl.abort(7); return; }
        // a short last group re-processes its last frame in the unused positions (the block holds nothing there)
#pragma unroll
        for (int q = 1; q < kGS; ++q) oth[q] = (q < g) ? oth[q] : oth[q - 1];
        // posterior of the frame: softmax of (own state + other side's state); both are stored relative to offsets
        // that keep each frame's largest term near 1, so no max-shift -- a normaliser outside [2^-100, 2^100] aborts
        // w = 2^(own + other) with own = arg + log2 s: as s * 2^(arg + other), no logarithm (|log2 s| <= 100 was just checked)
        R w[kGS], ex[kGS];
#pragma unroll
        for (int q = 0; q < kGS; ++q) {
            ex[q] = Num<R>::exp2(act ? ag[q] + oth[q] : NINF);
            w[q] = sg[q] * ex[q];
        }
        // the next own group's block of the other side
#ifdef ASG_PROBE_TAIL
        tl[2] = clock64();
#endif
        // the next own group's block of the other side, straight into `oth` (its last use in this iteration is above):
        // nothing waits for these loads until the next iteration's exponentials, a whole round of the consumers away
        load_other(n + kNC * kGS, oth);
#ifdef ASG_PROBE
        const long long seg_b = clock64();
#endif
        // frames 0-3: normalisers, rows, u, first MFMA batch; then frames 4-7 while those MFMAs run
        R u[kGS];
        auto half_group = [&](const int q0) -> bool {
            R z0 = w[q0], z1 = w[q0 + 1], z2 = w[q0 + 2], z3 = w[q0 + 3];
            wave_allsum4(z0, z1, z2, z3);
            const R Z[4] = {z0, z1, z2, z3};
            unsigned zlo = 0xffffffffu, zhi = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned zb = Rng<R>::bits(Z[q]);
                zlo = min(zlo, zb);
                zhi = max(zhi, zb);
            }
            if (zlo < Rng<R>::lo || zhi > Rng<R>::hi) return false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // u = posterior / s = 2^(arg + other) / Z (correctly rounded: its products with v = the recursion's vector sum
                // over the frames and must not be biased), posterior = s * u
                const R uq = div_nr(ex[q0 + q], Z[q]);
                const R post = sg[q0 + q] * uq;
                const int m = n + min(q0 + q, g - 1);
                if (q0 + q < g) buf_store_out(post * gscale, rs_g, voffg, (unsigned) frame(m) * grow_bytes);
                // xi: alpha side skips its first assembled index (the beta side's last one covers that transition)
                const bool take = q0 + q < g && (BETA || n + q0 + q > h);
                u[q0 + q] = (take && act) ? uq : R(0);
                pg[q0 + q] = act ? pg[q0 + q] : R(0);
            }
            float ua[4] = {u[q0], u[q0 + 1], u[q0 + 2], u[q0 + 3]}, va[4] = {pg[q0], pg[q0 + 1], pg[q0 + 2], pg[q0 + 3]};
            outer4_accumulate_f64<NT>(ua, va, acc);
            return true;
        };
        if (!half_group(0)) { ctl.abort(8); return; }
#ifdef ASG_PROBE
        const long long seg_c = clock64();
#endif
#ifdef ASG_PROBE_TAIL
        tl[3] = clock64();
#endif
        if (g > 4 && !half_group(4)) { ctl.abort(9); return; }
#ifdef ASG_PROBE_TAIL
        tl[4] = clock64();
        if (b == 0 && !BETA && lane == 0) { long long *d = (long long *) W.dbg + cw * 8; for (int k = 0; k < 5; ++k) d[k] = tl[k]; d[5] = n; }
#endif
#ifdef ASG_PROBE
        const long long seg_d = clock64();
#endif
#ifdef ASG_PROBE
        prb_seg[0] += seg_b - seg_a; prb_seg[1] += seg_c - seg_b; prb_seg[2] += seg_d - seg_c; prb_seg[3] += clock64() - seg_d; prb_seg[4] += 1;
#endif
        sv = sg[kGS - 1];
        n += kNC * kGS;
    }
    // (a consumer without a last group still has to let the producer's bookkeeping see "everything taken")
    lds_store_rlx(&L.cd[cw], len + kNC * kGS);
    // this wavefront's sums into its tile
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * r + (lane >> 4) + 4 * q, j = 16 * c + (lane & 15);
                sx[i][j] = acc[r * NT + c][q];
            }
    // ---- end of the chain: score (beta side), by the consumer that took the last group; as the three-wavefront kernel
    if (((((len - h + kGS - 1) / kGS) - 1) % kNC) != cw) return;
    {
        int spins = 0;
        while (!(lds_load_acq(&L.main_done) && lds_load_acq(&L.prod_done))) {
            if (L.stop()) return;
            if (++spins > kCapLds) { ctl.abort(10); return; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (BETA) {
        const int csum = lds_load_rlx(&L.csum);
        const double zsum = L.zsum;
        const R vlast = sv * lds_ldf(&L.e[(len - 1) & (kFR - 1)][lane]);
        const R sm = wave_allsum(act ? vlast : R(0));
        const unsigned smb = Rng<R>::bits(sm);
        if (!(smb >= Rng<R>::lo && smb <= Rng<R>::hi)) { ctl.abort(11); return; }
        score_out2 = zsum + (double) csum + (double) Num<R>::log2(sm);
    }
    PRB_END(W.dbg, BETA ? 3 : 2)
#ifdef ASG_PROBE
    if (b == 0 && lane == 0) { long long *d = (long long *) W.dbg + (BETA ? 9 : 8) * 5; for (int k = 0; k < 5; ++k) d[k] = prb_seg[k]; }
#endif
}

// 16 (GUARD: nsteps) steps of an aligned chain; the state of index m0 + k goes to ring slot (m0 + k - 1) & (kAR - 1) -- blocks
// start at m0 = 1 + 16 j, so the 16 slots of a block are consecutive and every ring store is base + constant -- and to HBM
// (every index, as the stand-alone chains do: a per-step "first half only" test costs more than the bytes).
// Branch-free inside a full block so that consecutive steps overlap.
template <bool BETA, bool GUARD>
__device__ __forceinline__ void aligned_steps(const float (&cur)[kPF], int nsteps, int m0, int len, double H2, double Dx,
                                              double ebias, float *ringrow, __amdgpu_buffer_rsrc_t rs,
                                              unsigned voff, unsigned row_bytes, double &st) {
    typedef float R;
    const double L2Ed = 1.4426950408889634;
    const int f0 = BETA ? len - 1 - m0 : m0;             // frame of index m0
    const unsigned soff0 = (unsigned) f0 * row_bytes;
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        if (!GUARD || k < nsteps) {
            if (!BETA) {
                const double em = fma((double) cur[k], L2Ed, ebias);
                const double stay = st + H2;
                const double come = prev_lane_or_zero<double>(st) + Dx;
                st = fmax(em + lse2_acc<R>(stay, come), kLZd);
            } else {
                const double y = fmax(fma((double) cur[k], L2Ed, ebias) + st, kLZd);
                const double stay = y + H2;
                const double go = next_lane_or_zero<double>(y) + Dx;
                st = fmax(lse2_acc<R>(stay, go), kLZd);
            }
            const R v = to_state<R>(st);
            ringrow[k * 64] = v;
            buf_store(v, rs, voff, BETA ? soff0 - (unsigned) k * row_bytes : soff0 + (unsigned) k * row_bytes);
        }
    }
}

// ------------------------------------------------------------------ aligned chain
// The stand-alone aligned chains (asg_chains.h) with two changes: every state also goes into the `ar` ring for the
// finisher of this side, and only the first half goes to HBM (for the finisher of the other side).
template <bool BETA, bool BF16>
__device__ __forceinline__ void fused_aligned(const Problem &P, const State &W, int b, AliSide &L, AliSide &O, int len,
                                              int h, double &score_out2, void *aoff, UttSync *us) {
    typedef float R;
    const AliCtl ctl{&L, &O, us};
    const int lane = threadIdx.x & 63;
    const int T = P.T, S = P.S;
    const AlignedSetup<R> A = aligned_setup<R>(P, b, lane);
    // emission of this lane's label at frame offset `e` (elements): fp32, or bfloat16 widened
    const unsigned short *inh = (const unsigned short *) P.inputs + (int64_t) b * P.is1 + (int64_t) A.tgt * P.is2;
    auto emis = [&](int64_t e) -> R { if constexpr (BF16) return bf16_bits_to_float(inh[e]); else return A.in[e]; };
    const unsigned row_bytes = (unsigned) S * sizeof(R);
    __amdgpu_buffer_rsrc_t rs = make_rsrc((R *) (BETA ? W.bb : W.ab) + (int64_t) b * T * S, (unsigned) T * row_bytes);
    const unsigned voff = lane < S ? (unsigned) lane * sizeof(R) : kOobOffset;
    const double L2Ed = 1.4426950408889634, H2 = (double) A.H2, Dp = (double) A.Dprev, Dn = (double) A.Dnext;
    auto frame = [&](int m) { return BETA ? len - 1 - m : m; };
    score_out2 = -1e300;
    PRB_DECL
    double C = 0.0;
    // index 0
    double st;
    if (!BETA) st = (lane == 0) ? fmax(fma((double) emis(0), L2Ed, (double) A.ebias), kLZd) : kLZd;
    else st = (lane == A.ol - 1) ? 0.0 : kLZd;
    {
        const R v = to_state<R>(st);
        L.ar[kAR - 1][lane] = v;                         // slot (0 - 1) & (kAR - 1)
        buf_store(v, rs, voff, (unsigned) frame(0) * row_bytes);
    }
    bool told = false;
    auto tell_first_half = [&](int upto) {              // indices [0, upto) done
        if (!told && upto >= h) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_store_rel(&L.ast_done, h);
            told = true;
        }
    };
    lds_store_rlx(&L.ar_done, 1);
    tell_first_half(1);
    const int nst = len - 1;
    R cur[kPF], nxt[kPF];
    // step k of a block starting at `done` reads emission frame ef = (alpha) 1+done+k / (beta) len-1-done-k and produces
    // the state of index 1+done+k
#pragma unroll
    for (int k = 0; k < kPF; ++k) cur[k] = emis((int64_t) (BETA ? max(len - 1 - k, 0) : min(1 + k, len - 1)) * P.is0);
    R last_raw = cur[0];
    // block prologue: ring space, renormalisation, block scale, per-block offsets; returns the emission bias of the block
    auto block_begin = [&](int done, int nsteps, bool &ok) -> double {
        // ring space: the slots this block overwrites held indices m0-kAR .. m0-kAR+15, last needed (as "previous") by
        // index m0-kAR+16
        const int m0 = 1 + done;
        ok = true;
        PRB_WAIT(0, ok = wait_finished(m0 + 17 - kAR, ctl);)
        {
            const R m = wave_allmax((R) st);
            if (m > R(-1e29)) { st = fmax(st - (double) m, kLZd); C += (double) m; }
        }
        const R z = aligned_block_scale<R>(cur, nsteps, A.act, A.ol);
        // the state of index m0 + k is stored relative to  Cbase + z * (k + 1):  the finishers turn stored states back
        // into absolute log-scores with these two numbers per block (ring slot here, HBM for the other side)
        const int j = done / kPF;
        if (lane == 0) {
            L.cb[j & 3][0] = C;
            L.cb[j & 3][1] = (double) z;
            if (m0 < h) {
                double *o = (double *) aoff + (int64_t) j * 2;
                o[0] = C;
                o[1] = (double) z;
            }
        }
        C += (double) z * nsteps;
        return (double) A.ebias - (double) z;
    };
    auto block_end = [&](int done, int nsteps) {
        asm volatile("" ::: "memory");
        lds_store_rlx(&L.ar_done, 1 + done + nsteps);
        tell_first_half(1 + done + nsteps);
    };
    int done = 0;
    // full blocks in their own loop: branch-free bodies whose consecutive steps overlap (a shared body with a runtime
    // step count gets a branch per step)
    for (; done + kPF <= nst; done += kPF) {
#pragma unroll
        for (int k = 0; k < kPF; ++k)
            nxt[k] = emis((int64_t) (BETA ? max(len - 1 - (done + kPF + k), 0) : min(1 + done + kPF + k, len - 1)) * P.is0);
        bool ok;
        const double ebias = block_begin(done, kPF, ok);
        if (!ok) return;
        // (HBM gets the first half only -- what the other side's finishers read; later blocks store out of bounds = nowhere)
        aligned_steps<BETA, false>(cur, kPF, 1 + done, len, H2, BETA ? Dn : Dp, ebias, &L.ar[done & (kAR - 1)][lane], rs,
                                   1 + done < h ? voff : kOobOffset, row_bytes, st);
        block_end(done, kPF);
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
    }
    last_raw = cur[0];                       // beta, no remainder: the frame-0 emission is the next one in line
    if (done < nst) {
        const int r = nst - done;
        bool ok;
        const double ebias = block_begin(done, r, ok);
        if (!ok) return;
        aligned_steps<BETA, true>(cur, r, 1 + done, len, H2, BETA ? Dn : Dp, ebias, &L.ar[done & (kAR - 1)][lane], rs,
                                  1 + done < h ? voff : kOobOffset, row_bytes, st);
        block_end(done, r);
#pragma unroll
        for (int k = 1; k < kPF; ++k) last_raw = (k == r) ? cur[k] : last_raw;
    }
    tell_first_half(len);
    if (BETA) {
        // S_aligned = beta_0[0] + I~_0[0]   (force_aligned_lattice.cpp:316)
        const double y = fma((double) last_raw, L2Ed, (double) A.ebias) + st;
        const double y0 = readlane(y, 0);
        score_out2 = (A.ol >= 1) ? C + y0 : -1e300;
    }
    PRB_END(W.dbg, BETA ? 5 : 4)
}

// ------------------------------------------------------------------ finisher of the aligned workgroup
// Aligned posterior of every second-half frame of this side -> P2[b][side][quad][s][4] (for the backward launch, which
// subtracts them, scattered to labels, from the rows the full workgroups wrote)
// and the stay / arrive edge posteriors accumulated per target position.
//
// No per-frame normalisation: sum_s alpha_t(s) beta_t(s) is the SAME number for every frame -- the aligned score -- so it
// is measured once (first frame of this side's half, one max + one sum reduction) and every later posterior is
//   exp2(alpha_hat + beta_hat + (offset_alpha(t) + offset_beta(t) - score))
// with the per-frame offsets the two chains publish per block (the stored states are relative to them).  The reference
// normalises each frame with a softmax (force_aligned_lattice.cpp:164-166); the two agree to fp32 rounding of the states.
template <bool BETA>
__device__ __forceinline__ void fused_afin(const Problem &P, const State &W, const FusedArgs &F, int b, AliSide &L,
                                           AliSide &O, int len, int h, UttSync *us, const int fw) {
    typedef float R;
    const AliCtl ctl{&L, &O, us};
    const int lane = threadIdx.x & 63;
    const int T = P.T, S = P.S;
    const R LZ = Num<R>::logzero();
    const AlignedSetup<R> A = aligned_setup<R>(P, b, lane);
    const bool sl = lane < S;
    const R H2 = A.H2, Dprev = A.Dprev;
    const unsigned rbS = (unsigned) S * sizeof(R);
    __amdgpu_buffer_rsrc_t ro = make_rsrc((R *) (BETA ? W.ab : W.bb) + (int64_t) b * T * S, (unsigned) T * rbS);
    // P2 of this side: [quad = (index - h) / 4][position s][4 consecutive indices] -- one 16-byte write-through store per
    // lane per four frames (4-byte write-through stores cost a fabric write per lane)
    __amdgpu_buffer_rsrc_t rp = make_rsrc((R *) F.p2 + ((int64_t) b * 2 + (BETA ? 1 : 0)) * (T + 8) * S, (unsigned) (T + 8) * rbS);
    const unsigned vS = (unsigned) (sl ? lane : 0) * (unsigned) sizeof(R);
    const unsigned vQ = sl ? (unsigned) lane * 16u : kOobOffset;
    const int nblk = (T + kPF - 1) / kPF + 1;
    const double *ao = (const double *) F.aoff + ((int64_t) b * 2 + (BETA ? 0 : 1)) * nblk * 2;    // the OTHER side's
    auto frame = [&](int m) { return BETA ? len - 1 - m : m; };
    // offsets of stored states, branch-free (uniform addresses: broadcast LDS reads, all issued before one wait)
    auto off_own = [&](int m) -> double {            // this side's index m >= 2 (ring slots of the last 4 blocks)
        const int j = (m - 1) >> 4, k = (m - 1) & 15;
        const V2<double> c = *reinterpret_cast<const V2<double> *>(&L.cb[j & 3][0]);
        return c.x + c.y * (double) (k + 1);
    };
    auto off_oth = [&](int mo) -> double {           // the other side's index mo >= 0 (its first half)
        const int j = (mo - 1) >> 4, k = (mo - 1) & 15;         // mo = 0: entry 0 = {0, 0}
        const V2<double> c = *reinterpret_cast<const V2<double> *>(&L.ob[j + 1][0]);
        return c.x + c.y * (double) (k + 1);
    };
    // sum of stay-edge posteriors; sum of state posteriors of frames >= 1 (arrive = accS - accH).  In double: a position
    // the alignment dwells on collects ~100 posteriors of ~1 per finisher, and fp32 would round each add at 4e-6
    double accH = 0, accS = 0;
    PRB_DECL
    PRB_WAIT(0, if (!wait_ge(&O.ast_done, len - h, ctl)) return;)      // the other side's aligned first half is visible
    // its block offsets, once, into LDS (launch_fused_forward admits at most kMaxBlk - 1 blocks per half)
    {
        const int oblk = (len - h + kPF - 1) / kPF;
        for (int j = lane; j < kMaxBlk - 1; j += 64) {
            const int jc = min(j, max(oblk - 1, 0));
            const double c0 = __hip_atomic_load(ao + 2 * jc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double c1 = __hip_atomic_load(ao + 2 * jc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L.ob[j + 1][0] = c0;
            L.ob[j + 1][1] = c1;
        }
        if (lane == 0) { L.ob[0][0] = 0.0; L.ob[0][1] = 0.0; }
        __builtin_amdgcn_wave_barrier();
    }
    int n = h + fw * kGS;                 // this wavefront's groups: fw, fw + kAF, ...
    R oth[kGS];
#pragma unroll
    for (int q = 0; q < kGS; ++q) oth[q] = buf_load<R>(ro, vS, (unsigned) max(frame(min(n + q, len - 1)), 0) * rbS);
    double Sd = 0.0;               // the aligned score (log2 units) as measured on this side's first frame
    bool feasible = false, calibrated = false;
    while (n < len) {
        const int g = min(kGS, len - n);
        PRB_WAIT(1, if (!wait_ge(&L.ar_done, n + g, ctl)) return;)
        R own[kGS], ownp[kGS];
        double K[kGS];
#pragma unroll
        for (int q = 0; q < kGS; ++q) {
            const int m = n + min(q, g - 1);
            own[q] = L.ar[(m - 1) & (kAR - 1)][lane];
            ownp[q] = L.ar[(m - 2) & (kAR - 1)][lane];        // alpha side: alpha-bar of the previous frame (m >= h >= 2)
            K[q] = off_own(m) + off_oth(len - 1 - m);
        }
        // the other side's state rows of the next group: plain (L1-cacheable) loads -- they were complete in L2 before
        // that side's ast_done, and this compute unit has not touched those lines before
        R othn[kGS];
#pragma unroll
        for (int q = 0; q < kGS; ++q) othn[q] = buf_load<R>(ro, vS, (unsigned) max(frame(min(n + kAF * kGS + q, len - 1)), 0) * rbS);
        // beta side: alpha-bar of the frame below the group's last frame (the next consecutive index, another wavefront's group)
        const R obelow = buf_load<R>(ro, vS, (unsigned) max(frame(min(n + g, len - 1)), 0) * rbS);
        PRB_WAIT(3, asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");)
        lds_store_rlx(&L.fd[fw], n + g + (kAF - 1) * kGS);
        if (!calibrated) {
            const R g0 = sl ? own[0] + oth[0] : LZ;
            const R mg = wave_allmax(g0);
            const R w0 = (mg > R(-1e29)) ? Num<R>::exp2(g0 - mg) : R(0);
            const R z0 = wave_allsum(w0);
            feasible = mg > R(-1e29) && z0 > R(0);                    // infeasible alignment -> no posterior anywhere
            Sd = feasible ? (double) mg + (double) Num<R>::log2(z0) + K[0] : 0.0;
            calibrated = true;
        }
        R p2v[kGS];
#pragma unroll
        for (int q = 0; q < kGS; ++q) {
            const R arg = (R) ((double) (own[q] + oth[q]) + (K[q] - Sd));
            p2v[q] = (feasible && sl) ? Num<R>::exp2(arg) : R(0);
        }
        // The offsets give the posterior up to 1 + O(1e-6) per position (fp32 rounding of the two stored states); over
        // hundreds of frames of a lattice whose edge posteriors must cancel against the full lattice's (tiny alphabets)
        // that shows.  Renormalise every frame over the positions, as the reference's softmax does
        // (force_aligned_lattice.cpp:164-166); a sum far from 1 can only be an infeasible or underflowed frame: left alone.
        {
            R z0 = p2v[0], z1 = p2v[1], z2 = p2v[2], z3 = p2v[3];
            wave_allsum4(z0, z1, z2, z3);
            R z4 = p2v[4], z5 = p2v[5], z6 = p2v[6], z7 = p2v[7];
            wave_allsum4(z4, z5, z6, z7);
            const R Z[kGS] = {z0, z1, z2, z3, z4, z5, z6, z7};
#pragma unroll
            for (int q = 0; q < kGS; ++q) p2v[q] *= (Z[q] > R(0.99) && Z[q] < R(1.01)) ? R(2) - Z[q] : R(1);    // 1/Z to 1e-8
        }
        {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const unsigned qoff = (unsigned) ((n - h) >> 2) * (unsigned) S * 16u;
            u4 a = {__float_as_uint(p2v[0]), __float_as_uint(p2v[1]), __float_as_uint(p2v[2]), __float_as_uint(p2v[3])};
            u4 c = {__float_as_uint(p2v[4]), __float_as_uint(p2v[5]), __float_as_uint(p2v[6]), __float_as_uint(p2v[7])};
            // (plain stores: nothing in THIS launch reads them)
            __builtin_amdgcn_raw_buffer_store_b128(a, rp, vQ, qoff, ASG_X_OUT_AUX);
            if (g > 4) __builtin_amdgcn_raw_buffer_store_b128(c, rp, vQ, qoff + (unsigned) S * 16u, ASG_X_OUT_AUX);
        }
#pragma unroll
        for (int q = 0; q < kGS; ++q) {
            if (q < g) {
                const int f = frame(n + q);
                const R post2 = p2v[q];
                if (f >= 1) {
                    // posterior of the STAY edge into (f, s) = post2 / (1 + 2^(arrive - stay)); needs alpha-bar of frame f-1
                    const R abprev = BETA ? ((q + 1 < g) ? oth[(q + 1) & (kGS - 1)] : obelow) : ownp[q];
                    const R ap = sl ? abprev : LZ;
                    const R d = (prev_lane_or_zero<R>(ap) + Dprev) - (ap + H2);
                    // (plain v_rcp: its bias only moves 1e-7 of the posterior between the stay and the arrive edge)
                    accH += (double) (post2 * Num<R>::rcp(R(1) + Num<R>::exp2(d)));
                    accS += (double) post2;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kGS; ++q) oth[q] = othn[q];
        n += kAF * kGS;
    }
    lds_store_rlx(&L.fd[fw], len + kAF * kGS);
    // edge posteriors of this side: [stay | arrive] per target position
    {
        double *ed = (double *) F.edges + (((int64_t) b * 2 + (BETA ? 1 : 0)) * kAF + fw) * 128;
        __hip_atomic_store(ed + lane, accH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ed + 64 + lane, accS - accH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (fw == 0) { PRB_END(W.dbg, BETA ? 7 : 6) }
}

// exact full-lattice score of one utterance by ONE wavefront (log-domain beta recursion, max-shifted log-sum-exps)
template <int NP>
__device__ __forceinline__ double slow_full_score(const Problem &P, int b, int len) {
    typedef float R;
    const int lane = threadIdx.x & 63;
    const int N = P.N;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    if (len < 1) return -1e300;
    const bool act = lane < N;
    const int lc = act ? lane : 0;
    const R *tcol = (const R *) P.transition + (int64_t) lc * P.ts1;
    const R *in = (const R *) P.inputs + (int64_t) b * P.is1 + (int64_t) lc * P.is2;
    ChainState<R> r;
    r.v = act ? R(0) : NINF;
    r.C = 0.0;
    if (len >= 2) r = slow_full_steps<R, true>(in, P.is0, tcol, P.ts0, N, lane, len - 1, len - 1, r.v, r.C, (R *) nullptr, 0, false);
    const R y = act ? fma(in[0], L2E, r.v) : NINF;
    const R my = fmax(wave_allmax(y), Num<R>::logzero());
    const R sm = wave_allsum(Num<R>::exp2(y - my));
    return r.C + (double) my + (double) Num<R>::log2(sm);
}

// ------------------------------------------------------------------ the fused forward kernel
template <int NP>
__device__ __forceinline__ void aligned_workgroup(int b, FusedShared<NP> &SH) {
    typedef float R;
    const Problem P = ld_problem(kernarg_params());
    const FusedArgs F = ld_fargs(kernarg_params());
    AliSide &LA = SH.u.h.A, &LB = SH.u.h.B;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = P.N, T = P.T;
    const int len = __builtin_amdgcn_readfirstlane(P.in_len ? clampi(P.in_len[b], 0, T) : T);
    const int mid = crossing(len);
    const bool fused = len >= kMinFused;
    UttSync *us = reinterpret_cast<UttSync *>(F.sync + 64) + b;
    if (threadIdx.x == 0) {
        LA.kill = 0; LA.ast_done = 0; LA.ar_done = 0;
        LB.kill = 0; LB.ast_done = 0; LB.ar_done = 0;
        for (int k = 0; k < kAF; ++k) { LA.fd[k] = mid + k * kGS; LB.fd[k] = len - mid + k * kGS; }
        SH.score_ali = -1e300;
    }
    __syncthreads();
    if (wave == 4 && b == 0) {
        // normalised transition rows for the exact stand-alone code (asg_assemble.h reads them).  BEFORE the roles: this wavefront is a
        // finisher, idle until the chains cross; behind its role (round 4) these ~3 k cycles sat between the end of utterance 0's
        // aligned workgroup and the word its two full workgroups wait for -- 4 us of the launch at T = 150 (tools/fused_flags.py)
        const State W = ld_state(kernarg_params());
        const bool act = lane < N;
        const int lc = act ? lane : 0;
        V2<R> e2[NP / 2];
        R Ri;
        load_norm_row<R, NP>((const R *) P.transition + (int64_t) lc * P.ts0, P.ts1, N, act, e2, Ri);
        if (act) {
            V2<R> *erow = reinterpret_cast<V2<R> *>((R *) W.ehat + (int64_t) lane * W.npad);
#pragma unroll
            for (int j = 0; j < NP / 2; ++j) erow[j] = e2[j];
            ((R *) W.rmax)[lane] = Ri;
        }
    }
    double sc2 = -1e300;
    if (fused) {
        const Problem P = ld_problem(kernarg_params());        // per role: see FusedParams
        const State W = ld_state(kernarg_params());
        const FusedArgs F = ld_fargs(kernarg_params());
        switch (wave) {
            case 0:
                __builtin_amdgcn_s_setprio(3);
                if (P.in_bf16) fused_aligned<false, true>(P, W, b, LA, LB, len, mid, sc2, (double *) F.aoff + ((int64_t) b * 2 + 0) * ((T + kPF - 1) / kPF + 1) * 2, us);
                else fused_aligned<false, false>(P, W, b, LA, LB, len, mid, sc2, (double *) F.aoff + ((int64_t) b * 2 + 0) * ((T + kPF - 1) / kPF + 1) * 2, us);
                break;
            case 1:
                __builtin_amdgcn_s_setprio(3);
                if (P.in_bf16) fused_aligned<true, true>(P, W, b, LB, LA, len, len - mid, sc2, (double *) F.aoff + ((int64_t) b * 2 + 1) * ((T + kPF - 1) / kPF + 1) * 2, us);
                else fused_aligned<true, false>(P, W, b, LB, LA, len, len - mid, sc2, (double *) F.aoff + ((int64_t) b * 2 + 1) * ((T + kPF - 1) / kPF + 1) * 2, us);
                break;
            // two finishers of a side on a SIMD of their own pair, the third beside the OTHER side's chain (which has priority)
            case 2: case 6: fused_afin<false>(P, W, F, b, LA, LB, len, mid, us, (wave - 2) >> 2); break;
            case 3: case 7: fused_afin<true>(P, W, F, b, LB, LA, len, len - mid, us, (wave - 3) >> 2); break;
            case 5: fused_afin<false>(P, W, F, b, LA, LB, len, mid, us, 2); break;
            case 4: fused_afin<true>(P, W, F, b, LB, LA, len, len - mid, us, 2); break;
            default: break;
        }
        if (wave == 1 && lane == 0) SH.score_ali = sc2;
    }
    if (wave == 5 && b == 0 && lane != 1) F.ticket2[lane] = 0;        // for the backward launch (word 1: see the closing workgroup)
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool gave_up = !fused || LA.stop() || LB.stop();
        // Everything another workgroup reads from this one (P2, the edge posteriors, this score) is stored write-through and
        // drained by its writer: NO release fence here -- it would write back the whole L2 (the ~100 KB of aligned states
        // this workgroup left dirty there: measured 5-6 us, on the path of the full workgroups' epilogue).
        __hip_atomic_store((double *) F.ascore + b, SH.score_ali, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned expect = 0u;
        if (!__hip_atomic_compare_exchange_strong(&us->adone, &expect, gave_up ? 2u : 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT)) {
            // the full workgroups closed this utterance without us (kClosedWithoutAligned): whatever this workgroup left in
            // the utterance's words goes back to zero, the claim last
            unsigned *w = reinterpret_cast<unsigned *>(us);
            for (int k = (int) (sizeof(UttSync) / sizeof(unsigned)) - 1; k >= 0; --k)
                __hip_atomic_store(w + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef ASG_PROBE
        if (b == 0) ((long long *) ld_state(kernarg_params()).dbg)[55] = clock64();
#endif
    }
}

// One direction of the fully-connected lattice of utterance b.
template <int NP, bool BETA>
__device__ __forceinline__ void full_workgroup(int b, FusedShared<NP> &SH) {
    typedef float R;
    constexpr int kNC = Consumers<NP>::n;
    const Problem P = ld_problem(kernarg_params());
    const FusedArgs F = ld_fargs(kernarg_params());
    FusedSide &L = SH.u.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = P.N, T = P.T;
    const int len = __builtin_amdgcn_readfirstlane(P.in_len ? clampi(P.in_len[b], 0, T) : T);
    const int mid = crossing(len);
    const int h = BETA ? len - mid : mid;
    const bool fused = len >= kMinFused;
    UttSync *us = reinterpret_cast<UttSync *>(F.sync + 64) + b;

    if (threadIdx.x == 0) {
        L.e_prod = 0; L.csum = 0; L.main_done = 0; L.prod_done = 0; L.kill = 0;
        for (int k = 0; k < kMaxNC; ++k) L.cd[k] = k == 0 ? 0 : 1 << 30;      // (slots >= kNC stay "infinitely far")
        L.st_done = 0;
        SH.score_full = -1e300;
        SH.adone = 0;
        SH.last = 0;
        SH.tr_ready = 0;
    }
#ifdef ASG_PROBE
    const long long ep_t0 = clock64();
    if (b == 0 && !BETA && threadIdx.x == 0) ((long long *) ld_state(kernarg_params()).dbg)[50] = ep_t0;
#endif
    for (int q = threadIdx.x; q < kFR * 64; q += kFusedThreads) (&L.s[0][0])[q] = __uint_as_float(kSentinel);
    TileLds<NP> &TL = SH.t;
    // (the consumers' tiles need no zeroing: each writes every element of its own as its last act)
    __syncthreads();

    double sc2 = -1e300;
#ifdef ASG_PROBE
    if (b == 0 && !BETA && threadIdx.x == 0) ((long long *) ld_state(kernarg_params()).dbg)[51] = clock64();
#endif

    // the transition matrix into LDS, by the six wavefronts that have nothing to do yet (the recursion wavefront and the
    // producer go ahead: the producer's first emission loads overlap with these)
    double est = 0, ear = 0;          // ... and (wavefront 4, while the roles run) the aligned edge posteriors of this side
    // ---- phase 1: the roles.  Control flow is uniform per wavefront; nothing in here uses a workgroup barrier.
    // Idle wavefronts go straight to the barrier: a waiting wavefront issues nothing.
    if (!BETA && wave == 7) {
        // padded frames get exactly-zero gradients (the reference: roll_to_end + masked softmax, utils.cpp:11-66)
        if (P.in_bf16) {
            unsigned short *gh = (unsigned short *) F.grad_inputs + (int64_t) b * N;
            if (lane < N) for (int t = len; t < T; ++t) gh[(int64_t) t * P.B * N + lane] = 0;
        } else {
            __amdgpu_buffer_rsrc_t rs_g = make_rsrc((R *) F.grad_inputs + (int64_t) b * N,
                                                    (unsigned) ((int64_t) (T - 1) * P.B * N + N) * (unsigned) sizeof(R));
            const unsigned voff = lane < N ? (unsigned) lane * sizeof(R) : kOobOffset;
            const unsigned grow_bytes = (unsigned) P.B * N * sizeof(R);
            for (int t = len; t < T; ++t) buf_store(R(0), rs_g, voff, (unsigned) t * grow_bytes);
        }
    }
    constexpr int kFill = 6;                                   // wavefronts 2 .. 7
    if (wave >= 2) {
        const R *tr = (const R *) P.transition;
        for (int k = (int) threadIdx.x - 128; k < N * N; k += 64 * kFill) {
            const int i = k / N, j = k - i * N;
            SH.trl[k] = tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&SH.tr_ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (fused) {
        const Problem P = ld_problem(kernarg_params());        // per role: see FusedParams
        const State W = ld_state(kernarg_params());
        const FusedArgs F = ld_fargs(kernarg_params());
        switch (wave) {
            case 0: __builtin_amdgcn_s_setprio(3); fused_main<NP, BETA>(P, b, L, us, len, W.dbg, SH.trl, &SH.tr_ready, kFill); break;
            case 1:
                if (P.in_bf16) duo_producer<NP, BETA, FusedSide, true>(P, b, L, SH.trl, &SH.tr_ready, kFill);
                else duo_producer<NP, BETA, FusedSide, false>(P, b, L, SH.trl, &SH.tr_ready, kFill);
                break;
            case 2: fused_consumer<NP, BETA>(P, W, F, b, L, us, len, h, TL.sx[0], sc2, 0); break;
            case 3: fused_consumer<NP, BETA>(P, W, F, b, L, us, len, h, TL.sx[1], sc2, 1); break;
            // (consumers on three SIMDs: the third beside the producer, which is light)
            case 5: if (kNC > 2) fused_consumer<NP, BETA>(P, W, F, b, L, us, len, h, TL.sx[2 % kNC], sc2, 2); break;
            case 6: if (kNC > 3) fused_consumer<NP, BETA>(P, W, F, b, L, us, len, h, TL.sx[3 % kNC], sc2, 3); break;
            case 4: {
                // the aligned workgroup's verdict and the edge posteriors of THIS side's frames (it finishes a little
                // before the recursion does; slow polls beside the recursion wavefront)
                unsigned v = 0;
                int spins = 0;
                while ((v = __hip_atomic_load(&us->adone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                    if (L.stop() || ++spins > kCapVerdictEarly) break;
                    __builtin_amdgcn_s_sleep(32);
                }
                v = __builtin_amdgcn_readfirstlane(v);
#ifdef ASG_PROBE
                if (b == 0 && !BETA && lane == 0) { ((long long *) W.dbg)[57] = clock64(); ((long long *) W.dbg)[58] = spins; }
#endif
                if (v == 1) {
                    const double *ed = (const double *) F.edges + ((int64_t) b * 2 + (BETA ? 1 : 0)) * kAF * 128;
                    for (int k = 0; k < kAF; ++k) {          // fixed order
                        est += __hip_atomic_load(ed + k * 128 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ear += __hip_atomic_load(ed + k * 128 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (v != 0 && lane == 0) SH.adone = (int) v;
#ifdef ASG_PROBE
                if (b == 0 && !BETA && lane == 0) ((long long *) W.dbg)[56] = clock64();
#endif
                break;
            }
            default: break;
        }
        if (BETA && lane == 0 && sc2 > -1e299) SH.score_full = sc2;
    }
#ifdef ASG_PROBE
    if (b == 0 && !BETA && lane == 0 && wave >= 2 && wave <= 6) ((long long *) ld_state(kernarg_params()).dbg)[57 + wave] = clock64();
#endif
#ifdef ASG_PROBE_TAIL
    if (b == 0 && !BETA && lane == 0 && wave >= 2 && wave <= 6) ((long long *) ld_state(kernarg_params()).dbg)[41 + wave] = clock64();
#endif
    // LDS-only barrier: what the epilogue reads from the roles is in LDS; the consumers' last row stores (a microsecond or
    // two from acknowledgement) need not have landed -- nothing in this launch reads them
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef ASG_PROBE
    if (b == 0 && !BETA && threadIdx.x == 0) ((long long *) ld_state(kernarg_params()).dbg)[52] = clock64();
#endif
    const bool own_trouble = !fused || L.stop();
    const R xx = L.x[lane & 63];                       // row / column maxima (the producer wrote them first thing)
    // the aligned workgroup's verdict, edge posteriors and score (it finishes about when this one does): the alpha
    // workgroup needs the edges for its tile; whoever closes the utterance needs the verdict
    auto wait_aligned = [&]() {
        if (threadIdx.x == 0 && SH.adone == 0) {
            unsigned v = 0;
            int spins = 0;
            while ((v = __hip_atomic_load(&us->adone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                if (++spins > kCapVerdict) {
                    // The aligned workgroup has not reported (it waits for nobody, so it has not been scheduled yet: a grid
                    // larger than the device, a co-running kernel).  Give up on it -- the utterance is flagged and redone
                    // exactly by the backward launch -- but CLAIM the word first: if the claim fails the verdict arrived
                    // after all; if it succeeds the aligned workgroup will find the claim when it finally runs and
                    // clears it itself (it is then the last one to touch this utterance's words).
                    unsigned expect = 0u;
                    v = __hip_atomic_compare_exchange_strong(&us->adone, &expect, kClosedWithoutAligned, __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            ? kClosedWithoutAligned : expect;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            SH.adone = (int) v;
        }
        __syncthreads();
    };
    wait_aligned();
#ifdef ASG_PROBE
    if (b == 0 && !BETA && threadIdx.x == 0) ((long long *) ld_state(kernarg_params()).dbg)[53] = clock64();
#endif
    // arrive NOW (the second full workgroup of the utterance will close it: score, loss, verdict, sync words): the tile
    // below is read by the next launch only, so the returning atomic's round trip overlaps with building it
    unsigned arrived = 0;
    if (threadIdx.x == 0) {
        if (BETA) __hip_atomic_store((double *) F.fscore + b, SH.score_full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (own_trouble) __hip_atomic_store(&us->kill, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrived = __hip_atomic_fetch_add(&us->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!own_trouble && SH.adone == 1) {
        // ---- phase 2: this side's share of the utterance's [N][N] tile
        EdgeLds<NP> &EL = SH.u.e;
        for (int k = threadIdx.x; k < NP * NP; k += kFusedThreads) EL.fxT[k] = 0;
        __syncthreads();
        if (wave == 4) {
            // aligned edge posteriors of THIS side's frames, scattered to [to][from]: with them the tile is a small
            // residual (full-lattice and aligned edge posteriors of the same frames nearly cancel for peaked lattices),
            // so the sum over tiles in the backward launch does not lose what the cancellation leaves
            const AlignedSetup<R> A = aligned_setup<R>(P, b, lane);
            const double stay = est, arrive = ear;
            if (A.act) {
                if (stay != 0.0) atomicAdd(&EL.fxT[A.tgt * N + A.tgt], (unsigned long long) __double2ll_rn(stay * Num<R>::kFix));
                if (lane >= 1 && arrive != 0.0) atomicAdd(&EL.fxT[A.tgt * N + A.prv], (unsigned long long) __double2ll_rn(arrive * Num<R>::kFix));
            }
        }
        // alpha: tile[i][j] = E[i][j] * sx[i][j] - aligned edges,   E = exp2(Tr2[i][j] - rowmax_i)
        // beta:  tile[i][j] = F[j][i] * sx[j][i] - aligned edges,   F = exp2(Tr2[i][j] - colmax_j)
        // (sx = the consumers' sums, added in the order 0, 1, ...)
        // one coalesced pass of the whole workgroup over the transition matrix
        const R L2E = Num<R>::log2e();
        R *tile_out = (R *) F.tiles + ((int64_t) b * 2 + (BETA ? 1 : 0)) * N * N;
        // (xx sits in a register of lane i; the tile loop wants it by index: back through LDS, behind the sums)
        R *xs = SH.xs;
        if (wave == 0) xs[lane] = xx;
        __syncthreads();
        for (int k = (int) threadIdx.x; k < N * N; k += kFusedThreads) {
            const int i = k / N, j = k - i * N;
            const R t2 = SH.trl[k] * L2E;
            double sum = 0;
#pragma unroll
            for (int c = 0; c < kNC; ++c) sum += BETA ? TL.sx[c][j][i] : TL.sx[c][i][j];
            // the difference in double: for peaked lattices it is a small residual of two sums of ~len / 2
            const double vd = (double) Num<R>::exp2(t2 - xs[BETA ? j : i]) * sum
                              - (double) (long long) EL.fxT[k] * (1.0 / Num<R>::kFix);
            tile_out[k] = (R) (vd * (double) F.gscale);
        }
    }
    // ---- phase 3: the SECOND full workgroup to have arrived closes the utterance
    if (threadIdx.x == 0) SH.last = arrived == 1u ? 1 : 0;
    __syncthreads();
#ifdef ASG_PROBE
    if (b == 0 && !BETA && threadIdx.x == 0) ((long long *) ld_state(kernarg_params()).dbg)[54] = clock64();
#endif
    if (!SH.last) return;
    wait_aligned();
    const bool flagged = !fused || SH.adone != 1 || __hip_atomic_load(&us->kill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (flagged) {
        // exact scores here (so that the loss of this launch is right), exact gradients in the backward launch
        const State W = ld_state(kernarg_params());
        Problem Q = P;
        if (P.in_bf16) {
            widen_utterance(P, F, b, kFusedThreads);
            __syncthreads();
            Q = problem_fp32_copy(P, F);
        }
        if (wave == 0) {
            const double sx = slow_full_score<NP>(Q, b, len);
            if (lane == 0) SH.score_full = sx;
        } else if (wave == 1) {
            // the stand-alone aligned beta chain, forward-only; its score lands in scores[B + b]
            FwdOut O{};
            O.aligned_scores = (R *) F.scores + P.B;
            aligned_beta_chain<R, false>(Q, W, O, b);
        }
        __syncthreads();
    }
    if (wave == 0) {
        const double fs = flagged ? SH.score_full : __hip_atomic_load((double *) F.fscore + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const R full = score_out<R>(fs);
        R ali;
        if (flagged) ali = __hip_atomic_load((R *) F.scores + P.B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else ali = score_out<R>(__hip_atomic_load((double *) F.ascore + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (lane == 0) {
            ((R *) F.scores)[b] = full;
            if (!flagged) ((R *) F.scores)[P.B + b] = ali;
            __hip_atomic_store(F.flags + b, flagged ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the cross-workgroup words of this utterance go back to zero (all three workgroups are done with them)
            // (word 0 = adone stays when it holds the claim of a time-out: the aligned workgroup clears it when it runs)
            unsigned *w = reinterpret_cast<unsigned *>(us);
            for (int k = SH.adone == (int) kClosedWithoutAligned ? 1 : 0; k < (int) (sizeof(UttSync) / sizeof(unsigned)); ++k)
                __hip_atomic_store(w + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the last utterance to close (fixed order inside): reduces the loss over the batch and counts the flagged
        // utterances for the backward launch (its reducers wait for exactly that many exact redos)
        R *lossb = (R *) F.dump;                    // [B] per-utterance losses for the reducing workgroup
        const R l = full - ali;
        if (F.reduction == 0 && lane == 0) ((R *) F.loss)[b] = l;
        unsigned ticket = 0;
        if (lane == 0) {
            __hip_atomic_store(lossb + b, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ticket = __hip_atomic_fetch_add(F.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket == (unsigned) (P.B - 1)) {
            double s = 0;
            int nf = 0;
            for (int q = lane; q < P.B; q += 64) {
                s += (double) __hip_atomic_load(lossb + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nf += __hip_atomic_load(F.flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1 : 0;
            }
            s = wave_allsum(s);
            nf = (int) wave_allsum((float) nf);          // exact: B < 2^24
            if (lane == 0) {
                if (F.reduction != 0) ((R *) F.loss)[0] = (R) (F.reduction == 2 ? s / P.B : s);
                F.ticket2[1] = (unsigned) nf;
                __hip_atomic_store(F.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// grid = 48 * ceil(B / 16): blocks come in groups of 48 = 16 utterances x {aligned, full alpha, full beta}; the three
// workgroups of utterance b = 16 G + 2 x + j (x < 8, j < 2) have indices 48 G + 24 j + {0, 8, 16} + x -- equal mod 8 (same
// XCD / L2 in practice) and close together in dispatch order (the alpha and beta workgroups wait for each other's first
// half).  Utterances 2x and 2x+1 share the XCD: their emission rows share 128-byte lines (a row is N floats), which
// otherwise every XCD fetches for itself.
template <int NP>
__global__ void __launch_bounds__(kFusedThreads, 2) fused_fwd_kernel(FusedParams KP) {
    __shared__ FusedShared<NP> SH;
    const int B = kernarg_params()->P.B;
    const int G = (int) blockIdx.x / 48, w = (int) blockIdx.x - 48 * G;
    const int j = w >= 24 ? 1 : 0, role = (w - 24 * j) >> 3;
#ifdef ASG_X_SPREAD_XCD
    const int b = 16 * G + 2 * ((w - 3 * role) & 7) + j;      // roles 0, 1, 2 of an utterance on XCDs x, x + 3, x + 6 (mod 8)
#else
    const int b = 16 * G + 2 * (w & 7) + j;
#endif
    if (b >= B) return;
#ifdef ASG_X_TEST_DELAY
    if ((b == 1 && role == 0) || (b == 2 && role == 1)) {
        if (threadIdx.x == 0) for (int q = 0; q < ASG_X_TEST_DELAY; ++q) __builtin_amdgcn_s_sleep(127);
        __syncthreads();
    }
#endif
#ifdef ASG_PROBE_XCC
    if (threadIdx.x == 0 && b < 16) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ((long long *) kernarg_params()->W.dbg)[b * 3 + role] = (long long) (xcc & 0xf);
    }
#endif
    if (role == 0) aligned_workgroup<NP>(b, SH);
    else if (role == 1) full_workgroup<NP, false>(b, SH);
    else full_workgroup<NP, true>(b, SH);
}

// ------------------------------------------------------------------ the backward kernel
// grid = kCH B + R workgroups of 256 threads.
//   workgroup kCH b + c:  utterance b.  Not flagged: the rows the forward launch left in grad_inputs hold gscale * (full
//                      posterior); finish them -- minus the aligned posteriors (P2) scattered to labels with deterministic
//                      fixed-point LDS adds, times the upstream gradient -- wave by wave, four frames (one P2 quad) at a
//                      time.  Flagged: workgroup c = 0 redoes the utterance exactly, scales, arrives.
//   workgroup kCH B + r:  once the redos have arrived, grad_transition[slice r] = sum_b g_b * (tile[b][alpha] + tile[b][beta])[slice r],
//                      tiles in ascending order.
constexpr int kBwdSlice = 64;
constexpr int kCH = 8;     // workgroups per utterance in the row pass
constexpr int kQB = 4;     // quads a wavefront has in flight
template <int NP>
__global__ void __launch_bounds__(256) fused_bwd_kernel(Problem P, State W, FusedArgs F) {
    typedef float R;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __shared__ AssembleLds<R, NP, 4> S;
    __shared__ __attribute__((aligned(16))) R lds4[4][64];
    __shared__ R part[4][kBwdSlice];
    __shared__ unsigned fxs[4][4][64];
    const int N = P.N, T = P.T, B = P.B;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Bp = B;
    if ((int) blockIdx.x < Bp * kCH) {
        // (mapping an utterance's blocks onto the XCD whose L2 the forward launch filled with its rows and posteriors was
        // tried: no measurable difference, 63.9 vs 63.5 us per step)
        const int b = (int) blockIdx.x / kCH, c = (int) blockIdx.x - b * kCH;
        // everything the row pass needs to know about its utterance in ONE round of loads, ahead of the flag test (behind it they
        // are a second and a third dependent memory round trip of a launch that is made of little else)
        const R g = ((const R *) F.grad_loss)[F.reduction == 0 ? b : 0];
        const int fl = F.flags[b];
        const int len_ld = P.in_len ? clampi(P.in_len[b], 0, T) : T;
        const int ol_ld = P.tg_len ? clampi(P.tg_len[b], 0, P.S) : P.S;
        const int64_t tg_ld = P.targets[(int64_t) b * P.gs0 + (int64_t) ((threadIdx.x & 63) < P.S ? (threadIdx.x & 63) : 0) * P.gs1];
        const bool flagged = fl != 0;
        if (flagged) {
            if (c != 0) return;
            FwdOut O{};
            O.full_scores = (R *) F.dump + B;            // scratch: the scores of this launch's forward stay as they are
            O.aligned_scores = (R *) F.dump + 2 * B;
            const Problem Q = P.in_bf16 ? problem_fp32_copy(P, F) : P;     // (widened by the forward launch's closing workgroup)
            if (wave == 0) full_alpha_chain<R, NP, 0, true>(Q, W, O, b, lds4[0]);
            else if (wave == 1) full_beta_chain<R, NP, 0, true>(Q, W, O, b, lds4[1]);
            else if (wave == 2) aligned_alpha_chain<R, true>(Q, W, O, b);
            else aligned_beta_chain<R, true>(Q, W, O, b);
            // the four chains' state rows are read back by all four wavefronts: make them visible past the L1
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            BwdArgs A{};
            A.unit_grad = 1;
            A.neg_aligned = 1;
            A.gscale = (double) F.gscale;
            A.grad_inputs = F.rows;
            A.chunk = T;
            A.nchunks = 1;
            assemble_frames<R, NP, 4>(Q, W, A, 3, b, 0, (R *) F.tiles + (int64_t) b * 2 * N * N, S);
            for (int k = threadIdx.x; k < N * N; k += 256) ((R *) F.tiles)[((int64_t) b * 2 + 1) * N * N + k] = R(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (g != R(1) || P.in_bf16) {
                R *gi = (R *) F.rows + (int64_t) b * N;
                unsigned short *gh = (unsigned short *) F.grad_inputs + (int64_t) b * N;
                const int total = T * N;
                for (int k = threadIdx.x; k < total; k += 256) {
                    const int t = k / N, i = k - t * N;
                    const R v = gi[(int64_t) t * B * N + i] * g;
                    if (P.in_bf16) gh[(int64_t) t * B * N + i] = float_to_bf16_bits(v);
                    else gi[(int64_t) t * B * N + i] = v;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(F.ticket2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // ---- the rows of utterance b
        const int Sx = P.S;
        const int len = __builtin_amdgcn_readfirstlane(len_ld);
        const int mid = crossing(len);
        // label of target position `lane` (clamped like aligned_setup); positions >= target length carry posterior 0:
        // give each its OWN word (all of them on one address would serialise the LDS add)
        const int tgt = lane < ol_ld ? clampi(tg_ld, 0, N - 1) : lane;
        const unsigned rbS = (unsigned) Sx * sizeof(R);
        __amdgpu_buffer_rsrc_t rg = make_rsrc((R *) F.rows + (int64_t) b * N,
                                              (unsigned) ((int64_t) (T - 1) * B * N + N) * (unsigned) sizeof(R));
        unsigned short *gh = (unsigned short *) F.grad_inputs + (int64_t) b * N;       // bfloat16 emissions: the output
        const unsigned voff = lane < N ? (unsigned) lane * sizeof(R) : kOobOffset;          // out of range: loads 0, stores nothing
        const unsigned vQ = lane < Sx ? (unsigned) lane * 16u : kOobOffset;
        const unsigned grow_bytes = (unsigned) B * N * sizeof(R);
        const R gscale = F.gscale;
        unsigned (*fx)[64] = fxs[wave];
#pragma unroll
        for (int r = 0; r < 4; ++r) fx[r][lane] = 0;
        // quads: the alpha side's (indices mid .. len-1 = frames, ascending), then the beta side's (indices len-mid .. len-1 =
        // frames mid-1 .. 0)
        const int nqa = (len - mid + 3) >> 2, nq = nqa + ((mid + 3) >> 2);
        for (int q0 = c * 4 + wave; q0 < nq; q0 += 4 * kCH * kQB) {
            u4 p2[kQB];
            R row[kQB][4];
            int fr[kQB][4], cnt[kQB];
#pragma unroll
            for (int u = 0; u < kQB; ++u) {
                const int q = q0 + u * 4 * kCH;
                const bool live = q < nq;
                const int side = (live && q >= nqa) ? 1 : 0, k = live ? (side ? q - nqa : q) : 0;
                const int h = side ? len - mid : mid;
                const int base = h + 4 * k;
                cnt[u] = live ? min(4, len - base) : 0;
                __amdgpu_buffer_rsrc_t rp = make_rsrc((R *) F.p2 + ((int64_t) b * 2 + side) * (T + 8) * Sx, (unsigned) (T + 8) * rbS);
                p2[u] = __builtin_amdgcn_raw_buffer_load_b128(rp, vQ, (unsigned) k * (unsigned) Sx * 16u, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = base + min(r, max(cnt[u] - 1, 0));
                    fr[u][r] = clampi(side ? len - 1 - m : m, 0, T - 1);
                    row[u][r] = buf_load<R>(rg, voff, (unsigned) fr[u][r] * grow_bytes);
                }
            }
#pragma unroll
            for (int u = 0; u < kQB; ++u) {
                if (cnt[u] > 0) {
                    const R pv[4] = {__uint_as_float(p2[u].x), __uint_as_float(p2[u].y), __uint_as_float(p2[u].z), __uint_as_float(p2[u].w)};
                    // scatter to labels: integer LDS adds commute -> repeated labels give bit-identical sums run to run
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(&fx[r][tgt], FrameFix<R>::to(r < cnt[u] ? pv[r] : R(0)));
                    __builtin_amdgcn_wave_barrier();
                    unsigned fv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) fv[r] = fx[r][lane];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 4; ++r) fx[r][lane] = 0;
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < cnt[u]) {
                            const R v = g * (row[u][r] - gscale * FrameFix<R>::from(fv[r]));
                            if (P.in_bf16) { if (lane < N) gh[(int64_t) fr[u][r] * B * N + lane] = float_to_bf16_bits(v); }
                            else buf_store_out(v, rg, voff, (unsigned) fr[u][r] * grow_bytes);
                        }
                }
            }
        }
        return;
    }
    // ---- reducers: wait for the exact redos of the flagged utterances (normally none: no wait at all)
    const int r = (int) blockIdx.x - Bp * kCH;
    {
        // (both words in one access: the count of flagged utterances was written by the forward launch, the arrivals of this one
        // normally stay 0)
        const unsigned long long tk = __hip_atomic_load((const unsigned long long *) F.ticket2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned nflag = (unsigned) (tk >> 32);
        int spins = 0;
        while ((spins == 0 ? (unsigned) tk : __hip_atomic_load(F.ticket2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < nflag) {
            if (++spins > (1 << 24)) break;               // cannot happen: arrivals never wait on anything
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const int n2 = N * N;
    const int k = min(r * kBwdSlice + lane, n2 - 1);
    const R *tiles = (const R *) F.tiles;
    const R *gl = (const R *) F.grad_loss;
    // wave w sums tiles w, w+4, ... (two per utterance; 32 loads in flight: ONE memory round trip up to 64 utterances),
    // then a fixed-order combine over the four waves
    constexpr int kIF = 32;
    R a[kIF];
#pragma unroll
    for (int q = 0; q < kIF; ++q) a[q] = 0;
    const int NTILE = 2 * B;
    for (int b0 = wave; b0 < NTILE; b0 += 4 * kIF) {
#pragma unroll
        for (int q = 0; q < kIF; ++q) {
            const int bb = b0 + 4 * q;
            const int bc = min(bb, NTILE - 1);
            const R v = __hip_atomic_load(tiles + (int64_t) bc * n2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const R g = gl[F.reduction == 0 ? (bc >> 1) : 0];
            a[q] += (bb < NTILE) ? v * g : R(0);
        }
    }
    R s = 0;
#pragma unroll
    for (int q = 0; q < kIF; ++q) s += a[q];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && r * kBwdSlice + lane < n2)
        ((R *) F.grad_transition)[k] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

template <int NP>
hipError_t launch_fused_np(const Problem &P, const State &W, const FusedArgs &F, bool backward, hipStream_t st) {
    if (!backward) {
        FusedParams KP{P, W, F};
        hipLaunchKernelGGL((fused_fwd_kernel<NP>), dim3(48 * ((P.B + 15) / 16)), dim3(kFusedThreads), 0, st, KP);
    } else {
        const int R = (P.N * P.N + kBwdSlice - 1) / kBwdSlice;
        hipLaunchKernelGGL((fused_bwd_kernel<NP>), dim3(P.B * kCH + R), dim3(256), 0, st, P, W, F);
    }
    return hipGetLastError();
}

hipError_t launch_fused(const Problem &P, const State &W, const FusedArgs &F, bool backward, hipStream_t stream) {
    const int N = P.N;
#ifdef ASG_DEV_ONLY_NP
    (void) N;
    return launch_fused_np<ASG_DEV_ONLY_NP>(P, W, F, backward, stream);
#else
    if (N <= 8) return launch_fused_np<8>(P, W, F, backward, stream);
    if (N <= 16) return launch_fused_np<16>(P, W, F, backward, stream);
    if (N <= 24) return launch_fused_np<24>(P, W, F, backward, stream);
    if (N <= 32) return launch_fused_np<32>(P, W, F, backward, stream);
    if (N <= 40) return launch_fused_np<40>(P, W, F, backward, stream);
    if (N <= 48) return launch_fused_np<48>(P, W, F, backward, stream);
    if (N <= 56) return launch_fused_np<56>(P, W, F, backward, stream);
    return launch_fused_np<64>(P, W, F, backward, stream);
#endif
}

}  // namespace

#ifdef ASG_PROBE
extern "C" void asg_dev_abort_codes(unsigned *out) {
    (void) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_abort_code), sizeof(unsigned) * 4);
    unsigned z[4] = {0, 0, 0, 0};
    (void) hipMemcpyToSymbol(HIP_SYMBOL(g_abort_code), z, sizeof(z));
}
#endif

hipError_t launch_fused_forward(const Problem &P, const State &W, const FusedArgs &F, hipStream_t stream) {
    // the recursion wavefronts address emission frames with 32-bit buffer offsets (as launch_fwd_small)
    const double fr = (double) (P.T - 1) * (double) P.is0 * sizeof(float), ln = 63.0 * (double) P.is2 * sizeof(float);
    if (P.is0 < 0 || P.is2 < 0 || fr >= 4294967296.0 || ln >= 2147483648.0) return hipErrorInvalidValue;
    return launch_fused(P, W, F, false, stream);
}

hipError_t launch_fused_backward(const Problem &P, const State &W, const FusedArgs &F, hipStream_t stream) {
    return launch_fused(P, W, F, true, stream);
}

}  // namespace asg
