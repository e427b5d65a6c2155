// torch_asg_amd/csrc/asg_chains.h -- gfx950 device code of the small-alphabet ASG recursions
// (N <= 64 labels, S <= 64 target positions): one recursion chain per workgroup (1 or 3 wavefronts).
//
// What each kernel replaces in the reference (paths under /root/reference/torch_asg/native/):
//   full_alpha_chain    fully_connected_lattice.cpp:9-29   (alpha recursion; no path_contrib tensor)
//   full_beta_chain     fully_connected_lattice.cpp:32-47,65-91 (beta recursion, length handling w/o roll_to_end)
//   aligned_alpha_chain force_aligned_lattice.cpp:84-111 + the gathers :15-82 / kernel.cu:7-226 fused in
//   aligned_beta_chain  force_aligned_lattice.cpp:113-154
//   bwd_small_kernel    fully_connected_lattice.cpp:49-63,93-105 + force_aligned_lattice.cpp:156-264,321-356
//                       (+ the atomicAdd scatter kernels force_aligned_lattice_kernel.cu:253-470)
//
// Design: the full-lattice recursion runs in the exp domain.  Lane i keeps row i of
// E = exp2(Tr2 - rowmax) in registers; the vector v_t (alpha_t = C_t + log2 v_t) is broadcast through LDS;
// s_i = sum_j E[i][j] v[j] (N FMAs); v_{t+1}[i] = s_i * e_{t+1}[i] with the emission factor
// e = exp2(I2 + rowmax - blockmax) prepared off the critical path.  Range control is a lagged power-of-two
// rescaling driven by lane N's row of ones (its row sum is the L1 norm of v): no reduction, exp or log on the
// dependent chain.  A sticky min/max of the row sums' bit patterns records any excursion outside 2^+-100; it is
// tested once per 16-step block and the block (or, on the three-wavefront path, the chain) is redone with exact
// max-shifted log-sum-exps, so the result is a true LSE for any input range.
// fp32 chains that have a compute unit to themselves run as THREE wavefronts (fwd_duo_kernel): the recursion
// wavefront keeps only the critical path, a producer prepares e_t, a consumer turns the row sums into the stored
// log-domain state and the score; they talk through LDS rings.  Everything else uses one wavefront per chain.
// The aligned lattice stays in the log domain (2-term LSE per node) because its band structure makes
// per-frame dynamic range unbounded for tight alignments.
//
// Things that mattered on gfx950 (each measured): unconditional clamped loads (a select
// around a load becomes a branch + vmcnt(0)); all LDS broadcast reads issued before the FMAs (sched_barrier);
// explicit vmcnt hints so the store queue is never drained at a loop head; DPP-fused reductions in inline asm;
// raw buffer stores whose bounds check replaces EXEC masking; no out-of-line calls in the hot kernel.
//
// This header holds the DEVICE code (recursion chains, the three-wavefront chain and its kernel); it is included by
// the translation units that instantiate kernels from it: asg_small_f32/f64.hip (recursion kernels),
// asg_fused.hip (recursions + gradient assembly in one launch).  Everything lives in an anonymous namespace.
#pragma once
#include "asg_common.h"
#include "asg_kernels.h"
#include <cstdlib>

namespace asg {

namespace {

constexpr int kPF = 16;     // emission prefetch depth (frames), register ring; also the unroll of one block
constexpr int kRenorm = 4;  // full lattice: subtract the frame max every kRenorm steps (must divide kPF)


__device__ __forceinline__ int clampi(int64_t v, int lo, int hi) {
    return v < lo ? lo : (v > hi ? hi : (int) v);
}

template <typename R> __device__ __forceinline__ V2<R> fma2(V2<R> a, V2<R> b, V2<R> c) {
    return __builtin_elementwise_fma(a, b, c);
}

// Row (or column) `lane` of the transition matrix in log2 units, max-normalised and exponentiated:
// e2[j] = exp2(Tr2[.] - mx).  All loads are unconditional (clamped indices) so they pipeline.
template <typename R, int NP>
__device__ __forceinline__ void load_norm_row(const R *base, int64_t stride, int N, bool act,
                                              V2<R> (&e2)[NP / 2], R &mx) {
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    R raw[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) raw[j] = base[(int64_t) (j < N ? j : 0) * stride];
    mx = NINF;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        raw[j] = (act && j < N) ? raw[j] * L2E : NINF;
        mx = fmax(mx, raw[j]);
    }
    if (mx == NINF) mx = 0;
#pragma unroll
    for (int j = 0; j < NP; j += 2) {
        e2[j / 2].x = Num<R>::exp2(raw[j] - mx);
        e2[j / 2].y = Num<R>::exp2(raw[j + 1] - mx);
    }
}

// s_i = sum_j e[j] * p_j with p_j living in lane j.
template <typename R, int NP, int MV>
__device__ __forceinline__ R matvec(const V2<R> (&e2)[NP / 2], R p, R *lds, int lane) {
    V2<R> a0 = {0, 0}, a1 = {0, 0};
    if (MV == 0) {
        // LDS broadcast: one ds_write_b32 per lane, then every lane reads the whole vector with
        // wide same-address (broadcast, conflict-free) reads.  A single wavefront owns `lds`, and the
        // LDS executes one wave's DS ops in order, so no s_barrier is needed -- only compiler ordering.
        lds[lane] = p;
        __builtin_amdgcn_wave_barrier();
        // Issue ALL broadcast reads back to back, then the FMAs (lgkmcnt(NP/4-1 .. 0) peels them off as
        // they land): one LDS round trip per step.  Left alone, hipcc's register-pressure scheduler
        // interleaves reads and FMAs 3-4 at a time and the step pays the LDS latency 3-4 times.
        V4<R> pv[NP / 4];
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) pv[j] = *reinterpret_cast<const V4<R> *>(lds + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) {
            a0 = fma2<R>(e2[2 * j], pv[j].xy, a0);
            a1 = fma2<R>(e2[2 * j + 1], pv[j].zw, a1);
        }
        __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
            V2<R> v0 = {readlane(p, j + 0), readlane(p, j + 1)};
            V2<R> v1 = {readlane(p, j + 2), readlane(p, j + 3)};
            a0 = fma2<R>(e2[j / 2], v0, a0);
            a1 = fma2<R>(e2[j / 2 + 1], v1, a1);
        }
    }
    V2<R> a = a0 + a1;
    return a.x + a.y;
}

// Exact log2-sum-exp2 over j of (trow[j*tstride]*log2e + v_j), v_j in lane j.  Rare path.
// Deliberately inlined (small rolled loops): an out-of-line call gives the kernel a stack (scratch),
// and a kernel that needs scratch pays ~20 us of extra dispatch cost on this stack.
template <typename R>
__device__ __forceinline__ R exact_lse_row(const R *trow, int64_t tstride, R v, int N, bool act) {
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    R mx = NINF;
    for (int j = 0; j < N; ++j) {
        R vj = readlane(v, j);
        R x = trow[j * tstride] * L2E + vj;
        mx = fmax(mx, (act && x == x) ? x : NINF);
    }
    R sm = 0;
    for (int j = 0; j < N; ++j) {
        R vj = readlane(v, j);
        R x = trow[j * tstride] * L2E + vj;
        sm += (mx == NINF || !act || x != x) ? R(0) : Num<R>::exp2(x - mx);
    }
    return (mx == NINF) ? NINF : mx + Num<R>::log2(sm);
}

template <typename R> __device__ __forceinline__ R score_out(double s2) {
    // log2-domain score -> natural log; anything at/below "log zero" is reported as -inf
    return (s2 < -1e29) ? Num<R>::ninf() : (R) (s2 * kLn2);
}

// Publish one pass's score; when the in-kernel loss reduction is on, the last of O.expected arriving passes
// reduces loss = full - aligned over the batch.  Hand-off without fences (MI355X guide, G16 "sc1 both sides"):
// scores are written with agent-scope (write-through) stores, drained with vmcnt(0), then a relaxed agent-scope
// ticket is drawn; the last arriver reads every score with agent-scope loads.  Placement-independent.
template <typename R>
__device__ __forceinline__ void publish_score(const FwdOut &O, R *slot, int b, int B, R score, int lane) {
    if (!O.loss) {
        if (lane == 0) slot[b] = score;
        return;
    }
    unsigned ticket = 0;
    if (lane == 0) {
        __hip_atomic_store(slot + b, score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket = __hip_atomic_fetch_add(O.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != (unsigned) (O.expected - 1)) return;
    const R *full = (const R *) O.full_scores, *ali = (const R *) O.aligned_scores;
    R *loss = (R *) O.loss;
    double s = 0;
    for (int q = lane; q < B; q += 64) {
        R f = __hip_atomic_load(full + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        R a = __hip_atomic_load(ali + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        R l = f - a;
        if (O.reduction == 0) loss[q] = l;
        s += (double) l;
    }
    if (O.reduction != 0) {
        s = wave_allsum(s);                      // fixed butterfly order: deterministic
        if (lane == 0) loss[0] = (R) (O.reduction == 2 ? s / B : s);
    }
    if (lane == 0) __hip_atomic_store(O.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ full lattice, alpha
// State per lane i: ah = alpha_t[i] in log2 units relative to the running offset C (double).
// Every kRenorm-th step the frame max is folded into C so that p = exp2(ah) stays in range; in between
// ah drifts by at most kRenorm-1 frames.
//
// Exactness guard without a branch on the critical path: every step ORs-in |log2 s| (as an unsigned bit
// pattern: finite < inf < NaN) into a per-lane sticky VGPR; after the 16-step block ONE test decides whether
// any row sum left the safe range (underflow, overflow, zero, NaN).  If so the chain rewinds to the state at
// block entry and redoes the block with `slow_full_steps` (exact max-shifted log-sum-exp per node), whose
// stores simply overwrite the fast attempt's.
template <typename R> struct ChainState { R v; double C; };

// Order-preserving 32-bit summary of a non-negative row sum (NaN and negative values sort above every finite one),
// the window [2^-lg_limit, 2^lg_limit] in the same encoding, and the binary exponent used for rescaling.
template <typename R> struct Rng;
template <> struct Rng<float> {
    static __device__ __forceinline__ unsigned bits(float s) { return __float_as_uint(s); }
    static constexpr unsigned lo = (127u - 100u) << 23, hi = (127u + 100u) << 23;
    static __device__ __forceinline__ int expo(float s) { return __builtin_amdgcn_frexp_expf(s); }
};
template <> struct Rng<double> {
    static __device__ __forceinline__ unsigned bits(double s) { return (unsigned) __double2hiint(s); }
    static constexpr unsigned lo = (1023u - 900u) << 20, hi = (1023u + 900u) << 20;
    static __device__ __forceinline__ int expo(double s) { return __builtin_amdgcn_frexp_exp(s); }
};

// The LDS broadcast of `matvec`, split in two so that independent work can be placed in the LDS latency window:
// bcast_issue writes the vector and issues every broadcast read; bcast_dot consumes them.
template <typename R, int NP>
__device__ __forceinline__ void bcast_issue(R v, R *lds, int lane, V4<R> (&pv)[NP / 4]) {
    lds[lane] = v;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NP / 4; ++j) pv[j] = *reinterpret_cast<const V4<R> *>(lds + 4 * j);
}
// A lone wavefront issues one instruction per 4 cycles whatever its kind, so an s_waitcnt per read costs as much as
// a v_pk_fma.  The reads are therefore awaited in groups of kWaitGroup (fp32: one ds_read_b128 per V4); the waits
// are explicit because s_waitcnt wants an immediate (hence the compile-time recursion).
#ifndef ASG_WAIT_GROUP
#define ASG_WAIT_GROUP 5
#endif
constexpr int kWaitGroup = ASG_WAIT_GROUP;
template <typename R, int NP, int J>
__device__ __forceinline__ void bcast_dot_step(const V2<R> (&e2)[NP / 2], const V4<R> (&pv)[NP / 4], V2<R> &a0, V2<R> &a1) {
    constexpr int NR = NP / 4;
    if constexpr (J < NR) {
        if constexpr (sizeof(R) == 4 && J % kWaitGroup == 0) {
            constexpr int last = (J + kWaitGroup - 1 < NR - 1) ? J + kWaitGroup - 1 : NR - 1;
            __builtin_amdgcn_sched_barrier(0);      // keep the FMAs of later groups behind their wait
            __builtin_amdgcn_s_waitcnt(0xC07F | ((NR - 1 - last) << 8));      // lgkmcnt only
            __builtin_amdgcn_sched_barrier(0);
        }
        a0 = fma2<R>(e2[2 * J], pv[J].xy, a0);
        a1 = fma2<R>(e2[2 * J + 1], pv[J].zw, a1);
        bcast_dot_step<R, NP, J + 1>(e2, pv, a0, a1);
    }
}
template <typename R, int NP>
__device__ __forceinline__ R bcast_dot(const V2<R> (&e2)[NP / 2], const V4<R> (&pv)[NP / 4]) {
    V2<R> a0 = {0, 0}, a1 = {0, 0};
    bcast_dot_step<R, NP, 0>(e2, pv, a0, a1);
    __builtin_amdgcn_wave_barrier();
    V2<R> a = a0 + a1;
    return a.x + a.y;
}

// Binary exponent of the L1 norm of the vector that went into the last mat-vec.  For N < 64 lane N carries a row
// of ones, so its row sum IS that norm (one v_frexp + one v_readlane); a full 64-label alphabet pays a reduction.
template <typename R, int NP>
__device__ __forceinline__ int scale_exponent(R s_prev, R vec, int N) {
    if (NP == 64 && N == 64) return Rng<R>::expo(wave_allsum(vec));
    return __builtin_amdgcn_readlane(Rng<R>::expo(s_prev), N);
}

template <typename R, int NP, int MV, bool STORE, bool GUARD>
__device__ __forceinline__ bool full_alpha_block(const R (&cur)[kPF], int nsteps, unsigned soff0, unsigned row_bytes,
                                                 const V2<R> (&e2)[NP / 2], R RiX, unsigned long long actmask, int N,
                                                 R *lds, int lane, __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                                 R &p, R &ah, double &C, __amdgpu_buffer_rsrc_t rsk, unsigned koff
#ifdef ASG_PROBE
                                                 , long long (&prb)[4]
#endif
                                                 ) {
    const R L2E = Num<R>::log2e();
    R kx = 0;      // lane k: the power-of-two exponent folded into frame k's emission factor (scale log of the gradient pass)
#ifdef ASG_PROBE
    const long long pc0 = clock64();
#endif
    // common scale of the block's emission factors: max over its frames and labels of I2 + rowmax
    R zl = cur[0];
#pragma unroll
    for (int k = 1; k < kPF; ++k)
        if (!GUARD || k < nsteps) zl = fmax(zl, cur[k]);
    const R zb = fmax(wave_allmax(fma(zl, L2E, RiX)), Num<R>::logzero());
    const R RiXz = RiX - zb;                             // -inf on lanes >= N
    R arg = fma(cur[0], L2E, RiXz), ee = Num<R>::exp2(arg);
#ifdef ASG_PROBE
    const long long pc1 = clock64();
#endif
    R s_prev = 1, arg_prev = 0;
    unsigned wlo = 0xffffffffu, whi = 0;
    int csum = 0;
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        if (!GUARD || k < nsteps) {
            V4<R> pv[NP / 4];
            if (MV == 0) bcast_issue<R, NP>(p, lds, lane, pv);
            __builtin_amdgcn_sched_barrier(0);
            // ---- LDS latency window: nothing here depends on the broadcast
            R arg_n = arg, ee_n = ee;
            if (k >= 1) {
                const unsigned sb = Rng<R>::bits(s_prev);
                wlo = min(wlo, sb);
                whi = max(whi, sb);
                if (STORE) buf_store(Num<R>::log2(s_prev) + arg_prev, rs, voff, soff0 + (unsigned) (k - 1) * row_bytes);
            }
            if (k + 1 < kPF) {
                arg_n = fma(cur[k + 1], L2E, RiXz);
                if ((k % kRenorm) == kRenorm - 2 && (!GUARD || k + 1 < nsteps)) {
                    const int ex = scale_exponent<R, NP>(s_prev, p, N);
                    arg_n -= (R) ex;
                    csum += ex;
                    if (STORE) kx = lane == k + 1 ? (R) ex : kx;
                }
                ee_n = Num<R>::exp2(arg_n);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- critical path: mat-vec, one multiply
            const R s = (MV == 0) ? bcast_dot<R, NP>(e2, pv) : matvec<R, NP, 1>(e2, p, lds, lane);
            p = s * ee;
            s_prev = s;
            arg_prev = arg;
            arg = arg_n;
            ee = ee_n;
        }
    }
#ifdef ASG_PROBE
    const long long pc2 = clock64();
    prb[0] += pc1 - pc0;
    prb[1] += pc2 - pc1;
#endif
    {
        const unsigned sb = Rng<R>::bits(s_prev);
        wlo = min(wlo, sb);
        whi = max(whi, sb);
        ah = Num<R>::log2(s_prev) + arg_prev;
        if (STORE) buf_store(ah, rs, voff, soff0 + (unsigned) ((GUARD ? nsteps : kPF) - 1) * row_bytes);
    }
    // scale log: frame k of the block was stored as log2(row sum) + I2 + Ri - zb - kx  (ScaleLog, asg_kernels.h)
    if (STORE) buf_store2(V2<R>{zb, kx}, rsk, lane < (GUARD ? nsteps : kPF) ? (unsigned) lane * (unsigned) (2 * sizeof(R)) : kOobOffset, koff);
    C += (double) zb * (double) (GUARD ? nsteps : kPF) + (double) csum;
    return (__ballot(wlo < Rng<R>::lo || whi > Rng<R>::hi) & actmask) != 0;
}

// Exact (slow, rare) recursion for `nsteps` frames starting at frame t_first: alpha or beta direction.
//   alpha (BETA=false): x_i = I2[t][i] + LSE_j(Tr2[i][j] + v_j),   v <- x - max, C += max
//   beta  (BETA=true):  y_j = I2[t][j] + v_j, C += max(y), y -= max, v_i <- LSE_j(Tr2[j][i] + y_j)   (writes frame t-1)
template <typename R, bool BETA>
__device__ __forceinline__ ChainState<R> slow_full_steps(const R *in, int64_t is0, const R *tline, int64_t tstride,
                                                      int N, int lane, int t_first, int nsteps, R v, double C,
                                                      R *out, int64_t out_stride, bool store) {
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const bool act = lane < N;
    for (int n = 0; n < nsteps; ++n) {
        const int t = BETA ? t_first - n : t_first + n;
        R em = act ? in[(int64_t) t * is0] * L2E : NINF;
        if (!BETA) {
            R x = em + exact_lse_row<R>(tline, tstride, v, N, act);
            R m = fmax(wave_allmax(x), LZ);
            v = x - m;
            C += (double) m;
            if (store && act) out[(int64_t) t * out_stride] = v;
        } else {
            R y = em + v;
            R m = fmax(wave_allmax(y), LZ);
            y -= m;
            C += (double) m;
            v = exact_lse_row<R>(tline, tstride, y, N, act);
            if (!act) v = NINF;
            if (store && act) out[(int64_t) (t - 1) * out_stride] = v;
        }
    }
    ChainState<R> r;
    r.v = v;
    r.C = C;
    return r;
}

template <typename R, int NP, int MV, bool STORE>
__device__ void full_alpha_chain(const Problem &P, const State &W, const FwdOut &O, int b, R *lds) {
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    const bool act = lane < N;
    const unsigned long long actmask = __ballot(act);
    const int lc = act ? lane : 0;
    const R *trow = (const R *) P.transition + (int64_t) lc * P.ts0;

    V2<R> e2[NP / 2];
    R Ri;
    load_norm_row<R, NP>(trow, P.ts1, N, act, e2, Ri);
    const R RiX = act ? Ri : NINF;
    if (STORE && !O.no_store && b == 0 && act) {
        // publish the normalised transition rows once per forward: the gradient-assembly kernel reads them
        // instead of redoing N loads + N exp2 in every one of its wavefronts
        V2<R> *erow = reinterpret_cast<V2<R> *>((R *) W.ehat + (int64_t) lane * W.npad);
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) erow[j] = e2[j];
        ((R *) W.rmax)[lane] = Ri;
    }
    if (lane == N) {                 // N < 64: lane N sums the broadcast vector (see scale_exponent)
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) e2[j] = V2<R>{1, 1};
    }

    const R *in = (const R *) P.inputs + (int64_t) b * P.is1 + (int64_t) lc * P.is2;
    // emission frames through buffer loads: lane offset in a VGPR, frame offset in an SGPR (launch_fwd_small checks
    // that both fit 32 bits); indices are clamped, so the resource needs no bound
    __amdgpu_buffer_rsrc_t rin = make_rsrc((R *) P.inputs + (int64_t) b * P.is1, 0xffffffffu);
    const unsigned vin = (unsigned) (lc * (int) P.is2) * (unsigned) sizeof(R);
    const unsigned fstride = (unsigned) P.is0 * (unsigned) sizeof(R);
    const unsigned row_bytes = (unsigned) N * sizeof(R);
    __amdgpu_buffer_rsrc_t rs = make_rsrc((R *) W.ah + (int64_t) b * T * N, (STORE && !O.no_store) ? (unsigned) T * row_bytes : 0u);
    const unsigned voff = act ? (unsigned) lane * sizeof(R) : kOobOffset;
    // scale log of this utterance ([T][2]; slot 0 = validity mark): see ScaleLog in asg_kernels.h
    constexpr unsigned kb = 2u * (unsigned) sizeof(R);
    __amdgpu_buffer_rsrc_t rsk = make_rsrc((R *) W.klog + (int64_t) b * T * 2, (STORE && !O.no_store) ? (unsigned) T * kb : 0u);
    auto void_log = [&](int t_first, int n) {      // frames redone by the exact code: no row-sum relation to log
        buf_store2(V2<R>{Num<R>::nan(), R(0)}, rsk, lane < n ? (unsigned) lane * kb : kOobOffset, (unsigned) t_first * kb);
    };

    double C = 0.0;
    R ah = NINF;
    if (len >= 1) {
        // frame 0: alpha_0 = I_0
        {
            R x = act ? in[0] * L2E : NINF;
            R m = fmax(wave_allmax(x), Num<R>::logzero());
            ah = x - m;
            C = (double) m;
            if (STORE) buf_store(ah, rs, voff, 0u);
            if (STORE) buf_store2(V2<R>{R(kScaleLogMark), R(0)}, rsk, lane == 0 ? 0u : kOobOffset, 0u);
        }
        // frames 1 .. len-1 in blocks of kPF, emissions prefetched one block ahead
        const int nst = len - 1;
        R cur[kPF], nxt[kPF];
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = buf_load<R>(rin, vin, (unsigned) min(1 + k, len - 1) * fstride);
        // enter the block loop with no load in flight, so the loop-head wait state is the steady-state one
        // (otherwise hipcc sizes the head-of-loop vmcnt for this first entry and drains the store queue every block)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        int done = 0;
        R p = Num<R>::exp2(ah);          // exp-domain state carried between blocks; `ah` is its exact log2 twin
#ifdef ASG_PROBE
        long long prb[4] = {0, 0, 0, 0};
        const long long pl0 = clock64();
#define ASG_PRB , prb
#else
#define ASG_PRB
#endif
        for (; done + kPF <= nst; done += kPF) {
#pragma unroll
            for (int k = 0; k < kPF; ++k)
                nxt[k] = buf_load<R>(rin, vin, (unsigned) min(1 + done + kPF + k, len - 1) * fstride);
            {
                const R ah0 = ah;
                const double C0 = C;
                const bool redo = full_alpha_block<R, NP, MV, STORE, false>(cur, kPF, (unsigned) (1 + done) * row_bytes,
                                                                            row_bytes, e2, RiX, actmask, N, lds, lane, rs,
                                                                            voff, p, ah, C, rsk, (unsigned) (1 + done) * kb ASG_PRB);
                // consume the prefetched frames BEFORE the (rare) branch: the load-completion wait is then an
                // exact vmcnt(#stores) here, instead of a conservative drain of the store queue after the merge
                // The 16 prefetch loads were issued before this block's 16 stores: "at most 16 VMEM ops outstanding"
                // == all loads have landed.  Saying so explicitly keeps hipcc from draining the store queue
                // (vmcnt(0/1)) at the loop head, where it has lost the exact count behind the branch below.
                __builtin_amdgcn_s_waitcnt(STORE ? 0x4F70 : 0x0F70);
#pragma unroll
                for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
                if (redo) {
                    ChainState<R> r = slow_full_steps<R, false>(in, P.is0, trow, P.ts1, N, lane, 1 + done, kPF, ah0, C0,
                                                                (R *) W.ah + (int64_t) b * T * N + lc, N, STORE && !O.no_store);
                    ah = r.v;
                    C = r.C;
                    p = Num<R>::exp2(ah);
                    if (STORE) void_log(1 + done, kPF);
                }
            }
        }
#ifdef ASG_PROBE
        if (b == 0 && lane == 0) {
            long long *d = (long long *) W.dbg;
            d[0] = 0x1234567890abcdefLL; d[1] = clock64() - pl0; d[2] = prb[0]; d[3] = prb[1]; d[4] = done / kPF;
        }
#endif
        if (done < nst) {
            const R ah0 = ah;
            const double C0 = C;
            if (full_alpha_block<R, NP, MV, STORE, true>(cur, nst - done, (unsigned) (1 + done) * row_bytes, row_bytes, e2,
                                                         RiX, actmask, N, lds, lane, rs, voff, p, ah, C, rsk, (unsigned) (1 + done) * kb ASG_PRB)) {
                ChainState<R> r = slow_full_steps<R, false>(in, P.is0, trow, P.ts1, N, lane, 1 + done, nst - done, ah0,
                                                            C0, (R *) W.ah + (int64_t) b * T * N + lc, N, STORE && !O.no_store);
                ah = r.v;
                C = r.C;
                if (STORE) void_log(1 + done, nst - done);
            }
        }
    }
    if (O.full_scores_alpha) {
        R mx = fmax(wave_allmax(ah), Num<R>::logzero());
        R sm = wave_allsum(Num<R>::exp2(ah - mx));
        double sc = (len >= 1) ? C + (double) mx + (double) Num<R>::log2(sm) : -1e300;
        if (lane == 0) ((R *) O.full_scores_alpha)[b] = score_out<R>(sc);
    }
}

// ------------------------------------------------------------------ full lattice, beta
// Iteration n handles frame t = len-1-n: y = I2[t] + bh[t]; it produces bh[t-1] = LSE_j(Tr[j][.] + y_j).
template <typename R, int NP, int MV, bool STORE, bool GUARD>
__device__ __forceinline__ bool full_beta_block(const R (&cur)[kPF], int nsteps, unsigned soff0, unsigned row_bytes,
                                                const V2<R> (&f2)[NP / 2], R CiX, unsigned long long actmask, int N,
                                                R *lds, int lane, __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                                R &q, R &bh, double &C) {
    const R L2E = Num<R>::log2e();
    R zl = cur[0];
#pragma unroll
    for (int k = 1; k < kPF; ++k)
        if (!GUARD || k < nsteps) zl = fmax(zl, cur[k]);
    const R zb = fmax(wave_allmax(fma(zl, L2E, CiX)), Num<R>::logzero());
    const R CiXz = CiX - zb;
    R ee = Num<R>::exp2(fma(cur[0], L2E, CiXz));
    R s_prev = 1, y = 0;
    unsigned wlo = 0xffffffffu, whi = 0;
    int csum = 0;
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        if (!GUARD || k < nsteps) {
            y = q * ee;
            V4<R> pv[NP / 4];
            if (MV == 0) bcast_issue<R, NP>(y, lds, lane, pv);
            __builtin_amdgcn_sched_barrier(0);
            R ee_n = ee;
            if (k >= 1) {
                const unsigned sb = Rng<R>::bits(s_prev);
                wlo = min(wlo, sb);
                whi = max(whi, sb);
                if (STORE) buf_store(CiX + Num<R>::log2(s_prev), rs, voff, soff0 - (unsigned) (k - 1) * row_bytes);
            }
            if (k + 1 < kPF) {
                R arg_n = fma(cur[k + 1], L2E, CiXz);
                if ((k % kRenorm) == kRenorm - 2 && (!GUARD || k + 1 < nsteps)) {
                    const int ex = scale_exponent<R, NP>(s_prev, y, N);
                    arg_n -= (R) ex;
                    csum += ex;
                }
                ee_n = Num<R>::exp2(arg_n);
            }
            __builtin_amdgcn_sched_barrier(0);
            const R s = (MV == 0) ? bcast_dot<R, NP>(f2, pv) : matvec<R, NP, 1>(f2, y, lds, lane);
            q = s;
            s_prev = s;
            ee = ee_n;
        }
    }
    {
        const unsigned sb = Rng<R>::bits(s_prev);
        wlo = min(wlo, sb);
        whi = max(whi, sb);
        bh = CiX + Num<R>::log2(s_prev);
        if (STORE) buf_store(bh, rs, voff, soff0 - (unsigned) ((GUARD ? nsteps : kPF) - 1) * row_bytes);
    }
    C += (double) zb * (double) (GUARD ? nsteps : kPF) + (double) csum;
    return (__ballot(wlo < Rng<R>::lo || whi > Rng<R>::hi) & actmask) != 0;
}

template <typename R, int NP, int MV, bool STORE>
__device__ void full_beta_chain(const Problem &P, const State &W, const FwdOut &O, int b, R *lds) {
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    const bool act = lane < N;
    const unsigned long long actmask = __ballot(act);
    const int lc = act ? lane : 0;
    const R *tcol = (const R *) P.transition + (int64_t) lc * P.ts1;      // column `lane`: Tr[j][lane]

    V2<R> f2[NP / 2];
    R Ci;
    load_norm_row<R, NP>(tcol, P.ts0, N, act, f2, Ci);
    const R CiX = act ? Ci : NINF;
    if (lane == N) {                 // see full_alpha_chain
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) f2[j] = V2<R>{1, 1};
    }

    const R *in = (const R *) P.inputs + (int64_t) b * P.is1 + (int64_t) lc * P.is2;
    __amdgpu_buffer_rsrc_t rin = make_rsrc((R *) P.inputs + (int64_t) b * P.is1, 0xffffffffu);     // see full_alpha_chain
    const unsigned vin = (unsigned) (lc * (int) P.is2) * (unsigned) sizeof(R);
    const unsigned fstride = (unsigned) P.is0 * (unsigned) sizeof(R);
    const unsigned row_bytes = (unsigned) N * sizeof(R);
    __amdgpu_buffer_rsrc_t rs = make_rsrc((R *) W.bh + (int64_t) b * T * N, (STORE && !O.no_store) ? (unsigned) T * row_bytes : 0u);
    const unsigned voff = act ? (unsigned) lane * sizeof(R) : kOobOffset;

    if (len < 1) {
        publish_score<R>(O, (R *) O.full_scores, b, P.B, NINF, lane);
        return;
    }
    double C = 0.0;
    R bh = act ? R(0) : NINF;
    if (STORE) buf_store(bh, rs, voff, (unsigned) (len - 1) * row_bytes);
    // steps n = 0 .. len-2 consume frames t = len-1 .. 1 and write bh[t-1]
    const int nst = len - 1;
    R cur[kPF], nxt[kPF];
#pragma unroll
    for (int k = 0; k < kPF; ++k) cur[k] = buf_load<R>(rin, vin, (unsigned) max(len - 1 - k, 0) * fstride);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // see full_alpha_chain
    int done = 0;
    // exp-domain state between blocks: beta = Ci + log2 q (relative to C); `bh` is its exact log2 twin
    R q = act ? Num<R>::exp2(bh - Ci) : R(0);
    for (; done + kPF <= nst; done += kPF) {
#pragma unroll
        for (int k = 0; k < kPF; ++k)
            nxt[k] = buf_load<R>(rin, vin, (unsigned) max(len - 1 - (done + kPF + k), 0) * fstride);
        {
            const R bh0 = bh;
            const double C0 = C;
            const bool redo = full_beta_block<R, NP, MV, STORE, false>(cur, kPF, (unsigned) (len - 2 - done) * row_bytes,
                                                                       row_bytes, f2, CiX, actmask, N, lds, lane, rs, voff,
                                                                       q, bh, C);
            __builtin_amdgcn_s_waitcnt(STORE ? 0x4F70 : 0x0F70);    // see full_alpha_chain
#pragma unroll
            for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
            if (redo) {
                ChainState<R> r = slow_full_steps<R, true>(in, P.is0, tcol, P.ts0, N, lane, len - 1 - done, kPF, bh0, C0,
                                                           (R *) W.bh + (int64_t) b * T * N + lc, N, STORE && !O.no_store);
                bh = r.v;
                C = r.C;
                q = act ? Num<R>::exp2(bh - Ci) : R(0);
            }
        }
    }
    R last_raw = cur[0];
    if (done < nst) {
        {
            const R bh0 = bh;
            const double C0 = C;
            if (full_beta_block<R, NP, MV, STORE, true>(cur, nst - done, (unsigned) (len - 2 - done) * row_bytes, row_bytes,
                                                        f2, CiX, actmask, N, lds, lane, rs, voff, q, bh, C)) {
                ChainState<R> r = slow_full_steps<R, true>(in, P.is0, tcol, P.ts0, N, lane, len - 1 - done, nst - done, bh0,
                                                           C0, (R *) W.bh + (int64_t) b * T * N + lc, N, STORE && !O.no_store);
                bh = r.v;
                C = r.C;
            }
        }
        // the frame-0 emission sits right after the last consumed ring slot
        const int r = nst - done;
        last_raw = cur[0];
#pragma unroll
        for (int k = 1; k < kPF; ++k) last_raw = (k == r) ? cur[k] : last_raw;
    }
    // frame 0: S_full = LSE_i(I_0[i] + beta_0[i])   (fully_connected_lattice.cpp:89)
    R y = fma(last_raw, L2E, bh);
    R my = fmax(wave_allmax(y), Num<R>::logzero());
    R sm = wave_allsum(Num<R>::exp2(y - my));
    publish_score<R>(O, (R *) O.full_scores, b, P.B, score_out<R>(C + (double) my + (double) Num<R>::log2(sm)), lane);
}

// ------------------------------------------------------------------ aligned lattice
template <typename R>
struct AlignedSetup {
    int len, ol, tgt, prv;
    bool act;
    R H2;      // Tr2[O_s][O_s]
    R Dprev;   // Tr2[O_s][O_{s-1}]   (edge s-1 -> s); log-zero for s == 0 and inactive lanes
    R Dnext;   // Tr2[O_{s+1}][O_s]   (edge s -> s+1); log-zero for s >= ol-1
    R ebias;   // 0 on active lanes, log-zero otherwise: em = fma(raw, log2e, ebias)
    const R *in;   // &inputs[0][b][O_s]
};

template <typename R>
__device__ __forceinline__ AlignedSetup<R> aligned_setup(const Problem &P, int b, int lane, bool valid = true) {
    AlignedSetup<R> A;
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    valid = valid && P.targets != nullptr;
    A.len = P.in_len ? clampi(P.in_len[b], 0, P.T) : P.T;
    A.ol = valid ? (P.tg_len ? clampi(P.tg_len[b], 0, P.S) : P.S) : 0;
    A.act = lane < A.ol;
    // unconditional, clamped loads (inactive lanes re-read position 0 / label 0)
    const int sc = A.act ? lane : 0;
    const int sp = (A.act && lane >= 1) ? lane - 1 : 0;
    const int sn = (lane + 1 < A.ol) ? lane + 1 : 0;
    int cur = 0, prv = 0, nxt = 0;
    if (valid) {
        const int64_t *tg = P.targets + (int64_t) b * P.gs0;
        cur = clampi(tg[(int64_t) sc * P.gs1], 0, P.N - 1);
        prv = clampi(tg[(int64_t) sp * P.gs1], 0, P.N - 1);
        nxt = clampi(tg[(int64_t) sn * P.gs1], 0, P.N - 1);
    }
    const R *tr = (const R *) P.transition;
    R h = tr[(int64_t) cur * P.ts0 + (int64_t) cur * P.ts1];
    R dp = tr[(int64_t) cur * P.ts0 + (int64_t) prv * P.ts1];
    R dn = tr[(int64_t) nxt * P.ts0 + (int64_t) cur * P.ts1];
    A.tgt = cur;
    A.prv = prv;
    A.H2 = A.act ? fmax(h * L2E, LZ) : R(0);
    A.Dprev = (A.act && lane >= 1) ? fmax(dp * L2E, LZ) : LZ;
    A.Dnext = (lane + 1 < A.ol) ? fmax(dn * L2E, LZ) : LZ;
    A.ebias = A.act ? R(0) : LZ;
    A.in = (const R *) P.inputs + (int64_t) b * P.is1 + (int64_t) cur * P.is2;
    return A;
}

// Common offset of a block's emissions (log2 units): taken out of the recursion and into the double offset C, so the
// log-domain state stays O(block spread) instead of growing by the emissions' absolute level every frame (fp32
// resolution at |x| ~ 1000 is 1e-4: emissions with a large common offset used to cost exactly that).  The offset is
// the MEAN over the block's frames and the utterance's target positions -- centring on the maximum would push
// zero-mean emissions down by ~3 per frame and cost precision there; the maximum is only the fallback when the mean
// is not finite (-inf emissions).
template <typename R>
__device__ __forceinline__ R aligned_block_scale(const R (&cur)[kPF], int nsteps, bool act, int ol) {
    R zs = 0, zl = Num<R>::ninf();
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        zs += (k < nsteps && act) ? cur[k] : R(0);
        zl = (k < nsteps && act) ? fmax(zl, cur[k]) : zl;
    }
    const R zmean = wave_allsum(zs) / (R) (nsteps * (ol > 0 ? ol : 1)) * Num<R>::log2e();
    if (zmean > R(-1e29) && zmean < R(1e29)) return zmean;
    const R zmax = wave_allmax(zl) * Num<R>::log2e();
    return (zmax > R(-1e29) && zmax < R(1e29)) ? zmax : R(0);
}

// The running state of the aligned chains is kept in DOUBLE even for fp32 problems: in the log domain every step adds
// numbers of magnitude ~30 (off-peak positions), so an fp32 state loses ~2e-6 per step and ~3e-5 over 400 frames --
// the whole error budget of the aligned gradient.  Only the bounded correction log2(1 + 2^d), d <= 0, is evaluated in
// the problem's precision (v_exp_f32 / v_log_f32); the sums are double adds.  Stores round to R once.
constexpr double kLZd = -1e30;
// (min - max = -|a - b| exactly: the modulus and the sign ride on the conversion and on v_exp as source modifiers.  Inside a
// block of frames the states are NOT clamped at log zero, only the emission term is (a -inf emission must not put -inf into a
// state: two of them side by side are inf - inf): a state falls below log zero by at most one -1e30 per frame until the block's
// renormalisation clamps it -- nowhere near the range of a double -- and every store clamps what it writes.)
template <typename R> __device__ __forceinline__ double lse2_acc(double a, double b) {
    const R d = (R) fabs(a - b);
    return fmax(a, b) + (double) Num<R>::log2(R(1) + Num<R>::exp2(-d));
}
template <typename R> __device__ __forceinline__ R to_state(double v) { return (R) fmax(v, kLZd); }

template <typename R, bool STORE, bool GUARD>
__device__ __forceinline__ void aligned_alpha_block(const R (&cur)[kPF], int nsteps, unsigned soff0, unsigned row_bytes,
                                                    const AlignedSetup<R> &A, double ebias, __amdgpu_buffer_rsrc_t rs,
                                                    unsigned voff, double &ab) {
    const double L2Ed = 1.4426950408889634, H2 = (double) A.H2, Dprev = (double) A.Dprev;
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        if (!GUARD || k < nsteps) {
            const double em = fmax(fma((double) cur[k], L2Ed, ebias), kLZd);
            const double stay = ab + H2;
            const double come = prev_lane_or_zero<double>(ab) + Dprev;      // lane 0: 0 + logzero
            ab = em + lse2_acc<R>(stay, come);
            if (STORE) buf_store(to_state<R>(ab), rs, voff, soff0 + (unsigned) k * row_bytes);
        }
    }
}

template <typename R, bool STORE>
__device__ void aligned_alpha_chain(const Problem &P, const State &W, const FwdOut &O, int b) {
    const int lane = threadIdx.x & 63;
    const int T = P.T, S = P.S;
    const AlignedSetup<R> A = aligned_setup<R>(P, b, lane);
    const int len = A.len;
    const unsigned row_bytes = (unsigned) S * sizeof(R);
    __amdgpu_buffer_rsrc_t rs = make_rsrc((R *) W.ab + (int64_t) b * T * S, (STORE && !O.no_store) ? (unsigned) T * row_bytes : 0u);
    const unsigned voff = lane < S ? (unsigned) lane * sizeof(R) : kOobOffset;

    if (STORE && !O.no_store && lane < S) {
        V2<R> u = {A.H2, A.Dprev};
        reinterpret_cast<V2<R> *>(W.asu)[(int64_t) b * S + lane] = u;
        int2 ii = {A.tgt, A.prv};
        reinterpret_cast<int2 *>(W.asi)[(int64_t) b * S + lane] = ii;
    }
    double C = 0.0;
    double ab = kLZd;
    if (len >= 1) {
        ab = (lane == 0) ? fmax(fma((double) A.in[0], 1.4426950408889634, (double) A.ebias), kLZd) : kLZd;
        if (STORE) buf_store(to_state<R>(ab), rs, voff, 0u);
        const int nst = len - 1;
        R cur[kPF], nxt[kPF];
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = A.in[(int64_t) min(1 + k, len - 1) * P.is0];
        int done = 0;
        for (; done + kPF <= nst; done += kPF) {
#pragma unroll
            for (int k = 0; k < kPF; ++k) nxt[k] = A.in[(int64_t) min(1 + done + kPF + k, len - 1) * P.is0];
            // renormalise once per block: the log domain is offset-free, this only bounds magnitudes
            R m = wave_allmax((R) ab);
            if (m > R(-1e29)) { ab = fmax(ab - (double) m, kLZd); C += (double) m; }
            const R z = aligned_block_scale<R>(cur, kPF, A.act, A.ol);
            C += (double) z * kPF;
            aligned_alpha_block<R, STORE, false>(cur, kPF, (unsigned) (1 + done) * row_bytes, row_bytes, A,
                                                 (double) A.ebias - (double) z, rs, voff, ab);
#pragma unroll
            for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
        }
        if (done < nst) {
            const R z = aligned_block_scale<R>(cur, nst - done, A.act, A.ol);
            C += (double) z * (nst - done);
            aligned_alpha_block<R, STORE, true>(cur, nst - done, (unsigned) (1 + done) * row_bytes, row_bytes, A,
                                                (double) A.ebias - (double) z, rs, voff, ab);
        }
    }
    if (O.aligned_scores_alpha) {
        const double last = (A.ol >= 1 && len >= 1) ? readlane(ab, A.ol - 1) : kLZd;
        if (lane == 0) ((R *) O.aligned_scores_alpha)[b] = score_out<R>(C + last);
    }
}

template <typename R, bool STORE, bool GUARD>
__device__ __forceinline__ void aligned_beta_block(const R (&cur)[kPF], int nsteps, unsigned soff0, unsigned row_bytes,
                                                   const AlignedSetup<R> &A, double ebias, __amdgpu_buffer_rsrc_t rs,
                                                   unsigned voff, double &bb) {
    const double L2Ed = 1.4426950408889634, H2 = (double) A.H2, Dnext = (double) A.Dnext;
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        if (!GUARD || k < nsteps) {
            const double y = fmax(fma((double) cur[k], L2Ed, ebias), kLZd) + bb;
            const double stay = y + H2;
            const double go = next_lane_or_zero<double>(y) + Dnext;
            bb = lse2_acc<R>(stay, go);
            if (STORE) buf_store(to_state<R>(bb), rs, voff, soff0 - (unsigned) k * row_bytes);
        }
    }
}

template <typename R, bool STORE>
__device__ void aligned_beta_chain(const Problem &P, const State &W, const FwdOut &O, int b) {
    const int lane = threadIdx.x & 63;
    const int T = P.T, S = P.S;
    const R NINF = Num<R>::ninf();
    const AlignedSetup<R> A = aligned_setup<R>(P, b, lane);
    const int len = A.len;
    const unsigned row_bytes = (unsigned) S * sizeof(R);
    __amdgpu_buffer_rsrc_t rs = make_rsrc((R *) W.bb + (int64_t) b * T * S, (STORE && !O.no_store) ? (unsigned) T * row_bytes : 0u);
    const unsigned voff = lane < S ? (unsigned) lane * sizeof(R) : kOobOffset;
    if (len < 1 || A.ol < 1) {
        publish_score<R>(O, (R *) O.aligned_scores, b, P.B, NINF, lane);
        return;
    }
    double C = 0.0;
    double bb = (lane == A.ol - 1) ? 0.0 : kLZd;
    if (STORE) buf_store(to_state<R>(bb), rs, voff, (unsigned) (len - 1) * row_bytes);
    const int nst = len - 1;
    R cur[kPF], nxt[kPF];
#pragma unroll
    for (int k = 0; k < kPF; ++k) cur[k] = A.in[(int64_t) max(len - 1 - k, 0) * P.is0];
    int done = 0;
    for (; done + kPF <= nst; done += kPF) {
#pragma unroll
        for (int k = 0; k < kPF; ++k) nxt[k] = A.in[(int64_t) max(len - 1 - (done + kPF + k), 0) * P.is0];
        R m = wave_allmax((R) bb);
        if (m > R(-1e29)) { bb = fmax(bb - (double) m, kLZd); C += (double) m; }
        const R z = aligned_block_scale<R>(cur, kPF, A.act, A.ol);
        C += (double) z * kPF;
        aligned_beta_block<R, STORE, false>(cur, kPF, (unsigned) (len - 2 - done) * row_bytes, row_bytes, A,
                                            (double) A.ebias - (double) z, rs, voff, bb);
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
    }
    R last_raw = cur[0];
    if (done < nst) {
        const R z = aligned_block_scale<R>(cur, nst - done, A.act, A.ol);
        C += (double) z * (nst - done);
        aligned_beta_block<R, STORE, true>(cur, nst - done, (unsigned) (len - 2 - done) * row_bytes, row_bytes, A,
                                           (double) A.ebias - (double) z, rs, voff, bb);
        const int r = nst - done;
#pragma unroll
        for (int k = 1; k < kPF; ++k) last_raw = (k == r) ? cur[k] : last_raw;
    }
    // S_aligned = beta_0[0] + I~_0[0]   (force_aligned_lattice.cpp:316)
    const double y = fma((double) last_raw, 1.4426950408889634, (double) A.ebias) + bb;
    const double y0 = readlane(y, 0);
    publish_score<R>(O, (R *) O.aligned_scores, b, P.B, score_out<R>(C + y0), lane);
}

// ------------------------------------------------------------------ aligned lattice, TWO utterances per wavefront (S <= 32)
// A short target leaves half of a wavefront's lanes idle (cfg 3 / cfg 4: 30 positions), and the double-precision steps of the
// aligned recursion cost the same for 30 active lanes as for 60.  For S <= 32 lanes 0-31 carry utterance 2 p and lanes 32-63
// utterance 2 p + 1, same direction: one instruction stream, half the wavefronts.  (Round 2 put the alpha AND beta chain of ONE
// utterance into a wavefront and lost: two directions make every frame index a per-lane quantity and leave one dependent
// chain where two wavefronts had two.  Two utterances of one direction share the frame index in the alpha loop; in the beta
// loop each half counts down from its own last frame, which costs two integer instructions per frame.)
// Formulas, stored states and scores are aligned_alpha_chain / aligned_beta_chain's, lane for lane; the neighbour exchange
// needs nothing new: position 0 has no arrive edge and position ol - 1 <= 31 no leave edge (log-zero weights), so what crosses
// the middle of the wavefront is absorbed.  Per-utterance lengths: the loop runs to the longer one, the shorter half is frozen
// (its state kept, its stores out of bounds).
template <typename R> __device__ __forceinline__ void half_reduce_prep_max(R &v);
template <> __device__ __forceinline__ void half_reduce_prep_max<float>(float &v) {
    asm volatile("s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 : "+v"(v));
}
template <> __device__ __forceinline__ void half_reduce_prep_max<double>(double &v) {
    v = fmax(v, dpp_mov<kDppXor1>(v, v));
    v = fmax(v, dpp_mov<kDppXor2>(v, v));
    v = fmax(v, dpp_mov<kDppHalfMirror>(v, v));
    v = fmax(v, dpp_mov<kDppMirror>(v, v));
}
template <typename R> __device__ __forceinline__ void half_reduce_prep_sum(R &v);
template <> __device__ __forceinline__ void half_reduce_prep_sum<float>(float &v) {
    asm volatile("s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n" "s_nop 1\n"
                 : "+v"(v));
}
template <> __device__ __forceinline__ void half_reduce_prep_sum<double>(double &v) {
    v += dpp_mov<kDppXor1>(v, v);
    v += dpp_mov<kDppXor2>(v, v);
    v += dpp_mov<kDppHalfMirror>(v, v);
    v += dpp_mov<kDppMirror>(v, v);
}
// maximum / sum over the lanes of this lane's HALF of the wavefront (lanes 0-31 | 32-63), in every lane of the half
template <typename R> __device__ __forceinline__ R half_allmax(R v, bool upper) {
    half_reduce_prep_max<R>(v);                  // every 16-lane row holds its own maximum
    const R lo = fmax(readlane(v, 0), readlane(v, 16)), hi = fmax(readlane(v, 32), readlane(v, 48));
    return upper ? hi : lo;
}
template <typename R> __device__ __forceinline__ R half_allsum(R v, bool upper) {
    half_reduce_prep_sum<R>(v);
    const R lo = readlane(v, 0) + readlane(v, 16), hi = readlane(v, 32) + readlane(v, 48);
    return upper ? hi : lo;
}

// Up to two scores at once (lanes 0 and 32 with `pub`): the wavefront draws its tickets with one atomic.
template <typename R>
__device__ __forceinline__ void publish_score_pair(const FwdOut &O, R *slot, int b, int B, R score, bool pub, int lane) {
    if (!O.loss) {
        if (pub) slot[b] = score;
        return;
    }
    const unsigned cnt = (unsigned) __popcll(__ballot(pub));
    if (cnt == 0) return;
    if (pub) __hip_atomic_store(slot + b, score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(O.counter, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket + cnt != (unsigned) O.expected) return;
    const R *full = (const R *) O.full_scores, *ali = (const R *) O.aligned_scores;
    R *loss = (R *) O.loss;
    double s = 0;
    for (int q = lane; q < B; q += 64) {
        R f = __hip_atomic_load(full + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        R a = __hip_atomic_load(ali + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        R l = f - a;
        if (O.reduction == 0) loss[q] = l;
        s += (double) l;
    }
    if (O.reduction != 0) {
        s = wave_allsum(s);
        if (lane == 0) loss[0] = (R) (O.reduction == 2 ? s / B : s);
    }
    if (lane == 0) __hip_atomic_store(O.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned kNoStore = 0xfffffff0u;       // voffset past any buffer (soffset 0): the store is dropped

// pair = index of the utterance pair (2 pair, 2 pair + 1).  Needs S <= 32 and 31-bit byte offsets into the emissions
// (aligned_pairs_apply in asg_small_impl.inc).
template <typename R, bool BETA>
__device__ void aligned_pair_chain(const Problem &P, const State &W, const FwdOut &O, int pair) {
    const int lane = threadIdx.x & 63, s = lane & 31;
    const bool upper = lane >= 32;
    const int T = P.T, S = P.S, B = P.B;
    const R NINF = Num<R>::ninf();
    const int b = 2 * pair + (upper ? 1 : 0);
    const bool uv = b < B;
    const int bc = uv ? b : B - 1;
    const AlignedSetup<R> A = aligned_setup<R>(P, bc, s, uv);
    const int len = uv ? A.len : 0, ol = A.ol;
    const int lmax = max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 32));
    const unsigned row_bytes = (unsigned) S * sizeof(R);
    const bool no_store = O.no_store != 0;
    __amdgpu_buffer_rsrc_t rs = make_rsrc(BETA ? W.bb : W.ab, no_store ? 0u : (unsigned) ((int64_t) B * T * S * sizeof(R)));
    const bool can_store = uv && s < S;
    const unsigned sbase = can_store ? (unsigned) ((((int64_t) b * T) * S + s) * sizeof(R)) : 0u;
    // emissions of this lane's label: one 32-bit byte offset per lane (signed arithmetic below: the launcher keeps it under 2^31)
    __amdgpu_buffer_rsrc_t rin = make_rsrc((void *) P.inputs, 0xffffffffu);
    const int fstride = (int) (P.is0 * (int64_t) sizeof(R));
    const int e0 = (int) (((int64_t) bc * P.is1 + (int64_t) A.tgt * P.is2) * (int64_t) sizeof(R));      // frame 0
    const int elast = e0 + (len >= 1 ? len - 1 : 0) * fstride;                                              // frame len - 1
    const double L2Ed = 1.4426950408889634, H2 = (double) A.H2, Dx = (double) (BETA ? A.Dnext : A.Dprev);

    if (!BETA && !no_store && uv && s < S) {
        V2<R> u = {A.H2, A.Dprev};
        reinterpret_cast<V2<R> *>(W.asu)[(int64_t) b * S + s] = u;
        int2 ii = {A.tgt, A.prv};
        reinterpret_cast<int2 *>(W.asi)[(int64_t) b * S + s] = ii;
    }
    R *score_slot = (R *) (BETA ? O.aligned_scores : O.aligned_scores_alpha);
    const bool dead = len < 1 || ol < 1;            // this half has no chain: score -inf
    double C = 0.0;
    double v = kLZd;
    // step n = 0 .. len - 2 of this lane's utterance; alpha: consumes frame n + 1 and writes it; beta: consumes frame
    // len - 1 - n and writes frame len - 2 - n
    const int nst = dead ? 0 : len - 1, nstmax = lmax >= 1 ? lmax - 1 : 0;
    int eoff;                                        // byte offset of the emission the next load takes
    unsigned soff;                                   // byte offset of the row the next step writes
    if (!BETA) {
        if (!dead) v = (s == 0) ? fmax(fma((double) buf_load<R>(rin, (unsigned) e0, 0u), L2Ed, (double) A.ebias), kLZd) : kLZd;
        buf_store(to_state<R>(v), rs, (dead || !can_store) ? kNoStore : sbase, 0u);
        eoff = min(e0 + fstride, elast);
        soff = sbase + row_bytes;
    } else {
        if (!dead) v = (s == ol - 1) ? 0.0 : kLZd;
        buf_store(to_state<R>(v), rs, (dead || !can_store) ? kNoStore : sbase + (unsigned) (len - 1) * row_bytes, 0u);
        eoff = elast;
        soff = sbase + (unsigned) (len >= 2 ? len - 2 : 0) * row_bytes;
    }
    R cur[kPF], nxt[kPF];
    auto fetch = [&](R (&dst)[kPF]) {
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            dst[k] = buf_load<R>(rin, (unsigned) eoff, 0u);
            eoff = BETA ? max(eoff - fstride, e0) : min(eoff + fstride, elast);
        }
    };
    fetch(cur);
    for (int done = 0; done < nstmax; done += kPF) {
        if (done + kPF < nstmax) fetch(nxt);
        // renormalise once per block: the log domain is offset-free, this only bounds magnitudes
        const R m = half_allmax<R>((R) v, upper);
        if (m > R(-1e29)) { v = fmax(v - (double) m, kLZd); C += (double) m; }
        // common offset of the block's emissions: their mean over this utterance's live frames and positions (see aligned_block_scale)
        R zs = 0, zl = NINF;
        int nlive = 0;
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            const bool live = done + k < nst && A.act;
            zs += live ? cur[k] : R(0);
            zl = live ? fmax(zl, cur[k]) : zl;
            nlive += (done + k < nst) ? 1 : 0;
        }
        R z = half_allsum<R>(zs, upper) / (R) (nlive * (ol > 0 ? ol : 1)) * Num<R>::log2e();
        if (!(z > R(-1e29) && z < R(1e29))) {
            const R zmax = half_allmax<R>(zl, upper) * Num<R>::log2e();
            z = (zmax > R(-1e29) && zmax < R(1e29)) ? zmax : R(0);
        }
        C += (double) z * (double) nlive;
        const double ebias = (double) A.ebias - (double) z;
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            if (done + k < nstmax) {                 // (wave-uniform)
                const bool live = done + k < nst;
                const double em = fmax(fma((double) cur[k], L2Ed, ebias), kLZd);
                double nv;
                if (!BETA) {
                    const double stay = v + H2;
                    const double come = prev_lane_or_zero<double>(v) + Dx;       // position 0: 0 + log-zero
                    nv = em + lse2_acc<R>(stay, come);
                } else {
                    const double y = em + v;
                    const double stay = y + H2;
                    const double go = next_lane_or_zero<double>(y) + Dx;         // position ol - 1: log-zero edge
                    nv = lse2_acc<R>(stay, go);
                }
                v = live ? nv : v;
                buf_store(to_state<R>(v), rs, (live && can_store) ? soff : kNoStore, 0u);
                soff = BETA ? soff - row_bytes : soff + row_bytes;
            }
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
    }
    if (!score_slot) return;
    double sc;
    if (!BETA) {
        // alpha: the state of position ol - 1 at the last frame
        const int at = (upper ? 32 : 0) + (ol >= 1 ? ol - 1 : 0);
        const double last = __shfl(v, at);
        sc = dead ? -1e300 : C + last;
    } else {
        // S_aligned = beta_0[0] + I~_0[0]   (force_aligned_lattice.cpp:316)
        const double y = fma((double) buf_load<R>(rin, (unsigned) e0, 0u), L2Ed, (double) A.ebias) + v;
        sc = dead ? -1e300 : C + y;                  // (lanes 0 and 32 hold position 0)
    }
    const R out = dead ? NINF : score_out<R>(sc);
    if (BETA) publish_score_pair<R>(O, score_slot, b, B, out, uv && s == 0, lane);
    else if (uv && s == 0) score_slot[b] = out;
}

// ------------------------------------------------------------------ full lattice, two wavefronts per chain (fp32)
// A lone wavefront issues one instruction every 4 cycles, whatever its kind: the recursion is bound by its
// instruction COUNT.  The duo path therefore leaves on the recursion wavefront ("main") only what is on the
// critical path -- scale, LDS broadcast, N FMAs -- and gives everything else to a second wavefront of the same
// workgroup ("helper", own SIMD, own issue slots): emission loads, the block scale and its wave reduction, the
// exp2 of the emission factors, the log2 + store of every frame's state, the range guard and the final score.
//
// Both directions run the same loop (X = row max Ri for alpha / column max Ci for beta, f_n = n or len-1-n):
//   v_n = s_{n-1} * e_n,   e_n = 2^(I2[f_n] + X - zb) [* 2^-ex],   s_n = M v_n,   s_{-1} = 2^-X
//   alpha: log2 alpha[f_n] = log2 s_{n-1} + arg_n      beta: log2 beta[f_n] = X + log2 s_{n-1}      (up to C)
//   score = sum(zb) + sum(ex) + log2 sum_i v_{len-1}[i]
// LDS rings (32 frames): helper -> main  e_n (and arg_n for the helper itself);  main -> helper  s_n.
// The s ring needs no counter: the helper resets a consumed slot to a NaN sentinel and polls the next one.
// The helper publishes how many frames of e it has produced; main checks that once per 16-frame block, which by
// construction also proves that the s slots it is about to overwrite have been consumed.
// A row sum outside the safe range (or a poll that never completes) makes the helper raise `verdict = 2`:
// main then redoes the whole chain with the single-wavefront code above, which has the exact fallback.
constexpr int kRing = 32;
constexpr unsigned kSentinel = 0x7fc0deadu;
constexpr int kSpinCap = 1 << 22;

struct DuoLds {
    __attribute__((aligned(16))) float p[64];
    float s[kRing][64];
    float e[kRing][64];
    float a[kRing][64];
    int e_prod;        // helper -> main: frames of e produced
    int verdict;       // helper -> main: 0 running, 1 done, 2 abort (redo with the exact single-wavefront chain)
    int csum;          // main -> helper: sum of the rescaling exponents
    int main_done;     // main -> helper
    int kill;          // main -> helpers: stop (spin cap hit)
    int c_done;        // consumer -> producer: last frame n whose s_{n-1} has been taken out of the ring
    int prod_done;     // producer -> consumer: zsum is final
    double zsum;       // producer -> consumer: sum of the block scales
    float x[64];       // producer -> consumer: row / column maxima
    float zb[4];       // producer -> consumer (alpha): block scales of the last four 16-frame blocks (scale log)
    // the buffer main broadcasts step n's vector through, and the helpers' common "stop" test
    static constexpr int kR = kRing;       // ring depth in frames (a multiple of 32)
    __device__ __forceinline__ float *pslot(int) { return p; }
    __device__ __forceinline__ int consumed() { return __hip_atomic_load(&c_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ bool stop() {
        return __hip_atomic_load(&verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 2 ||
               __hip_atomic_load(&kill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
    }
};

__device__ __forceinline__ int lds_load_acq(int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_load_rlx(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_rel(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_rlx(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ float lds_ldf(float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_stf(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the producer's block scale for the consumer's scale log (the fused step's rings have no such consumer)
template <class LdsT> __device__ __forceinline__ void post_block_scale(LdsT &, int, float) {}
__device__ __forceinline__ void post_block_scale(DuoLds &L, int K, float zb) { lds_stf(&L.zb[K & 3], zb); }

// mat-vec consumer with `EXTRA` younger LDS operations in flight behind the broadcast reads
// The main wavefront of the duo path is not issue-bound (about 40 instructions per 240-cycle step), so it waits for
// the broadcast reads in pairs: the FMAs then track the arrival of the data instead of starting after half of it.
#ifndef ASG_DUO_WAIT_GROUP
#define ASG_DUO_WAIT_GROUP 2
#endif
constexpr int kDuoWaitGroup = ASG_DUO_WAIT_GROUP;
template <int NP, int EXTRA, int J>
__device__ __forceinline__ void duo_dot_step(const V2<float> (&e2)[NP / 2], const V4<float> (&pv)[NP / 4], V2<float> &a0,
                                             V2<float> &a1) {
    constexpr int NR = NP / 4;
    if constexpr (J < NR) {
        if constexpr (J % kDuoWaitGroup == 0) {
            constexpr int last = (J + kDuoWaitGroup - 1 < NR - 1) ? J + kDuoWaitGroup - 1 : NR - 1;
            constexpr int cnt = (NR - 1 - last) + (last == NR - 1 ? 0 : EXTRA);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F | ((cnt > 15 ? 15 : cnt) << 8));
            __builtin_amdgcn_sched_barrier(0);
        }
        a0 = fma2<float>(e2[2 * J], pv[J].xy, a0);
        a1 = fma2<float>(e2[2 * J + 1], pv[J].zw, a1);
        duo_dot_step<NP, EXTRA, J + 1>(e2, pv, a0, a1);
    }
}

// 16 steps of the main wavefront: n = n0 .. n0+15 (GUARD: only the first `nsteps`).  Returns nothing: range
// checking is the helper's job.  s_prev enters as s_{n0-1} and leaves as the last row sums.
template <int NP, bool GUARD, class LdsT>
__device__ __forceinline__ void duo_main_block(LdsT &L, int n0, int nsteps, const V2<float> (&e2)[NP / 2], int N, int lane,
                                               float e_cur, float &s_prev, int &csum, int need_next, bool &next_ready,
                                               float &e_next_first) {
    constexpr int KR = LdsT::kR;
    const int half = (n0 & (KR - 16));             // ring position of this block (n0 is a multiple of 16)
    int ex = 0;
    int ep_early = 0;
    next_ready = false;
#pragma unroll
    for (int j = 0; j < kPF; ++j) {
        if (!GUARD || j < nsteps) {
            float v = s_prev * e_cur;
            if ((j % kRenorm) == kRenorm - 1) v = ldexpf(v, -ex);
            V4<float> pv[NP / 4];
            bcast_issue<float, NP>(v, L.pslot(half + j), lane, pv);
            // hand s_{n-1} to the helper (slot (n-1) & 31; nothing to hand over before the first step)
            if (j > 0) lds_stf(&L.s[(half + j - 1) & (KR - 1)][lane], s_prev);
            else if (n0 > 0) lds_stf(&L.s[(half + KR - 1) & (KR - 1)][lane], s_prev);
            float e_nxt = e_cur;
            if (j + 1 < kPF) e_nxt = lds_ldf(&L.e[half + j + 1][lane]);     // within this block's produced half
            // look ahead: is the NEXT block's e already there?  Asked at step 12, known at step 14, and then step 15
            // fetches that block's first e like any other -- the block boundary costs no LDS round trip
            if (!GUARD && j == kPF - 4) ep_early = lds_load_rlx(&L.e_prod);
            if (!GUARD && j == kPF - 2) next_ready = __builtin_amdgcn_readfirstlane(ep_early) >= need_next;
            if (!GUARD && j == kPF - 1) e_next_first = lds_ldf(&L.e[(half + 16) & (KR - 1)][lane]);
            __builtin_amdgcn_sched_barrier(0);
            if ((j % kRenorm) == kRenorm - 2 && (!GUARD || j + 1 < nsteps)) {
                ex = __builtin_amdgcn_readlane(Rng<float>::expo(s_prev), N);
                csum += ex;
            }
            V2<float> a0 = {0, 0}, a1 = {0, 0};
            __builtin_amdgcn_sched_barrier(0);
            duo_dot_step<NP, 2, 0>(e2, pv, a0, a1);
            __builtin_amdgcn_wave_barrier();
            V2<float> a = a0 + a1;
            s_prev = a.x + a.y;
            e_cur = e_nxt;
        }
    }
}

template <int NP, bool STORE, bool BETA>
__device__ __forceinline__ void duo_main(const Problem &P, const State &W, const FwdOut &O, int b, DuoLds &L) {
    typedef float R;
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const int len = __builtin_amdgcn_readfirstlane(P.in_len ? clampi(P.in_len[b], 0, T) : T);
    const bool act = lane < N;
    const int lc = act ? lane : 0;
    const R *tline = (const R *) P.transition + (int64_t) lc * (BETA ? P.ts1 : P.ts0);
    V2<R> e2[NP / 2];
    R X;
    load_norm_row<R, NP>(tline, BETA ? P.ts0 : P.ts1, N, act, e2, X);
    if (!BETA && STORE && !O.no_store && b == 0 && act) {
        V2<R> *erow = reinterpret_cast<V2<R> *>((R *) W.ehat + (int64_t) lane * W.npad);
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) erow[j] = e2[j];
        ((R *) W.rmax)[lane] = X;
    }
    if (lane == N) {
#pragma unroll
        for (int j = 0; j < NP / 2; ++j) e2[j] = V2<R>{1, 1};
    }
    bool redo = false;
    if (len >= 2) {
        const int nst = len - 1;                   // mat-vecs n = 0 .. len-2
        R s_prev = act ? Num<R>::exp2(-X) : R(0);
        int csum = 0;
        bool next_ready = false;
        R e_next_first = 0;
#ifdef ASG_PROBE
        long long pr_poll = 0, pr_blk = 0; const long long pr0 = clock64();
#endif
        for (int n0 = 0; n0 < nst && !redo; n0 += kPF) {
            const int nsteps = min(kPF, nst - n0);
#ifdef ASG_PROBE
            const long long pa = clock64();
#endif
            // e for frames n0 .. n0+nsteps (one beyond the last step is only needed by the helper itself)
            const int need = min(n0 + kPF, len);
            int spins = 0;
            R e_first = e_next_first;
            while (!next_ready) {
                // the counter, the verdict and the block's first e in ONE round trip: LDS executes a wavefront's
                // reads in order, so an e read issued behind a counter read that says "ready" sees produced data
                const int ep = lds_load_rlx(&L.e_prod);
                const int vd = lds_load_rlx(&L.verdict);
                e_first = lds_ldf(&L.e[n0 & 16][lane]);
                asm volatile("" ::: "memory");
                if (vd == 2) { redo = true; break; }
                if (ep >= need) break;
                if (++spins > kSpinCap) { lds_store_rlx(&L.kill, 1); redo = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (redo) break;
#ifdef ASG_PROBE
            const long long pb = clock64();
#endif
            const int need_next = min(n0 + 2 * kPF, len);
            if (nsteps == kPF)
                duo_main_block<NP, false>(L, n0, kPF, e2, N, lane, e_first, s_prev, csum, need_next, next_ready, e_next_first);
            else
                duo_main_block<NP, true>(L, n0, nsteps, e2, N, lane, e_first, s_prev, csum, need_next, next_ready, e_next_first);
#ifdef ASG_PROBE
            const long long pc = clock64();
            pr_poll += pb - pa; pr_blk += pc - pb;
#endif
        }
#ifdef ASG_PROBE
        if (!BETA && b == 0 && lane == 0) {
            long long *d = (long long *) W.dbg;
            d[0] = 0x1234567890abcdefLL; d[1] = clock64() - pr0; d[2] = pr_poll; d[3] = pr_blk; d[4] = (nst + kPF - 1) / kPF;
        }
#endif
        if (!redo) {
            lds_stf(&L.s[(nst - 1) & (kRing - 1)][lane], s_prev);
            lds_store_rlx(&L.csum, csum);
            lds_store_rel(&L.main_done, 1);
        }
    } else {
        lds_store_rlx(&L.csum, 0);
        lds_store_rel(&L.main_done, 1);
    }
    // wait for the helper's verdict
    if (!redo) {
        int spins = 0;
        while (true) {
            const int vd = lds_load_acq(&L.verdict);
            if (vd == 1) break;
            if (vd == 2) { redo = true; break; }
            if (++spins > kSpinCap) { lds_store_rlx(&L.kill, 1); redo = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
#ifdef ASG_PROBE
    if (lane == 0) ((long long *) W.dbg)[20 + (BETA ? 1 : 0)] = redo ? 7 : 5;
#endif
    if (redo) {
        // the helper has stopped (its stores are drained before it raises the verdict; after a kill it exits at
        // its next poll); wait for that, then redo the chain with the exact-fallback single-wavefront code
        int spins = 0;
        while (lds_load_acq(&L.verdict) != 2 && ++spins < kSpinCap) __builtin_amdgcn_s_sleep(2);
        if (BETA) full_beta_chain<R, NP, 0, STORE>(P, W, O, b, L.p);
        else full_alpha_chain<R, NP, 0, STORE>(P, W, O, b, L.p);
    }
}

// Producer wavefront: emission loads, block scale, e_n (and arg_n) into the rings.  Only loads on its VMEM queue.
// tr_lds / tr_ready (fused step): the transition matrix as a compact [N][N] copy that other wavefronts of the workgroup
// are filling in LDS; *tr_ready reaches tr_need once it is complete.  Waited for AFTER the first emission loads are out.
// bfloat16 -> float is a 16-bit shift (buffer_load_ushort + v_lshlrev)
__device__ __forceinline__ float bf16_bits_to_float(unsigned short h) { return __uint_as_float((unsigned) h << 16); }
template <bool BF16>
__device__ __forceinline__ float emis_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    if constexpr (BF16) return bf16_bits_to_float((unsigned short) __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0));
    else return buf_load<float>(rs, voff, soff);
}

template <int NP, bool BETA, class LdsT, bool BF16 = false>
__device__ __forceinline__ void duo_producer(const Problem &P, int b, LdsT &L, const float *tr_lds = nullptr,
                                             int *tr_ready = nullptr, int tr_need = 0) {
    typedef float R;
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const int len = __builtin_amdgcn_readfirstlane(P.in_len ? clampi(P.in_len[b], 0, T) : T);
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const bool act = lane < N;
    const int lc = act ? lane : 0;
    if (len < 1) return;
    constexpr unsigned ESZ = BF16 ? 2u : (unsigned) sizeof(R);
    __amdgpu_buffer_rsrc_t rin = make_rsrc((char *) P.inputs + (int64_t) b * P.is1 * ESZ, 0xffffffffu);
    const unsigned vin = (unsigned) (lc * (int) P.is2) * ESZ;
    const unsigned fstride = (unsigned) P.is0 * ESZ;
    const int nblk = (len + kPF - 1) / kPF;
    auto frame = [&](int n) { int nn = min(n, len - 1); return BETA ? len - 1 - nn : nn; };
    R blk0[kPF], nxt[kPF];
#pragma unroll
    for (int j = 0; j < kPF; ++j) blk0[j] = emis_load<BF16>(rin, vin, (unsigned) frame(j) * fstride);
#pragma unroll
    for (int j = 0; j < kPF; ++j) nxt[j] = emis_load<BF16>(rin, vin, (unsigned) frame(kPF + j) * fstride);
    // X = max over the row (alpha) / column (beta) of the transition matrix, log2 units
    R X = NINF;
    {
        const R *tline = (const R *) P.transition + (int64_t) lc * (BETA ? P.ts1 : P.ts0);
        int64_t ts = BETA ? P.ts0 : P.ts1;
        if (tr_lds) {
            int spins = 0;
            while (__hip_atomic_load(tr_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < tr_need) {
                if (L.stop() || ++spins > kSpinCap) return;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            tline = tr_lds + lc * (BETA ? 1 : N);
            ts = BETA ? N : 1;
        }
        R raw[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) raw[j] = tline[(int64_t) (j < N ? j : 0) * ts];
#pragma unroll
        for (int j = 0; j < NP; ++j) X = fmax(X, (act && j < N) ? raw[j] * L2E : NINF);
        if (X == NINF) X = 0;
    }
    const R XX = act ? X : NINF;
    L.x[lane] = XX;
    double zsum = 0.0;
    auto produce = [&](int K, const R (&em)[kPF]) {          // e (and arg) for frames n = 16K .. 16K+15, n < len
        const int nv = min(kPF, len - K * kPF);
        R zl = em[0];
#pragma unroll
        for (int j = 1; j < kPF; ++j) zl = (j < nv) ? fmax(zl, em[j]) : zl;
        const R zb = fmax(wave_allmax(fma(zl, L2E, XX)), LZ);
        const R Xz = XX - zb;
        const int half = (K * kPF) & (LdsT::kR - 1);
#pragma unroll
        for (int j = 0; j < kPF; ++j) {
            const R arg = fma(em[j], L2E, Xz);
            if (!BETA) lds_stf(&L.a[half + j][lane], arg);
            lds_stf(&L.e[half + j][lane], Num<R>::exp2(arg));
        }
        zsum += (double) zb * (double) nv;
        if (!BETA && lane == 0) post_block_scale(L, K, zb);
        // one wavefront's LDS operations execute in order: the counter lands after the values, no wait needed
        asm volatile("" ::: "memory");
        lds_store_rlx(&L.e_prod, min((K + 1) * kPF, len));
    };
    produce(0, blk0);
    if (nblk > 1) produce(1, nxt);
    constexpr int kAhead = LdsT::kR / kPF;          // blocks the ring holds
    for (int K = 2; K < nblk; ++K) {
#pragma unroll
        for (int j = 0; j < kPF; ++j) nxt[j] = emis_load<BF16>(rin, vin, (unsigned) frame(K * kPF + j) * fstride);
        // block K reuses the ring slots of block K - kAhead: wait until the consumer has seen s_{16(K-kAhead+1)-1},
        // i.e. main has finished block K - kAhead   (kAhead = 2: the consumer has seen s_{16(K-1)-1})
        int spins = 0;
        while (L.consumed() < (K - kAhead + 1) * kPF) {
            if (L.stop() || ++spins > kSpinCap) return;
            __builtin_amdgcn_s_sleep(2);
        }
        produce(K, nxt);
    }
    L.zsum = zsum;
    asm volatile("" ::: "memory");
    lds_store_rlx(&L.prod_done, 1);
}

// Consumer wavefront: range guard, log2 + store of every frame's state, final score.  Only stores on its VMEM queue.
template <int NP, bool STORE, bool BETA>
__device__ __forceinline__ void duo_consumer(const Problem &P, const State &W, const FwdOut &O, int b, DuoLds &L) {
    typedef float R;
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const int len = __builtin_amdgcn_readfirstlane(P.in_len ? clampi(P.in_len[b], 0, T) : T);
    const R NINF = Num<R>::ninf();
    const bool act = lane < N;
    const unsigned long long actmask = __ballot(act);
    if (len < 1) {
        if (BETA) publish_score<R>(O, (R *) O.full_scores, b, P.B, NINF, lane);
        else if (O.full_scores_alpha && lane == 0) ((R *) O.full_scores_alpha)[b] = NINF;
        lds_store_rel(&L.verdict, 1);
        return;
    }
    const unsigned row_bytes = (unsigned) N * sizeof(R);
    __amdgpu_buffer_rsrc_t rs = make_rsrc((R *) (BETA ? W.bh : W.ah) + (int64_t) b * T * N, (STORE && !O.no_store) ? (unsigned) T * row_bytes : 0u);
    const unsigned voff = act ? (unsigned) lane * sizeof(R) : kOobOffset;
    // alpha: scale log of the stored states (ScaleLog, asg_kernels.h).  This chain stores log2 s_{n-1} + arg_n and applies the
    // rescale exponent to the VECTOR of step n (n % 4 == 3, exponent of lane N of s_{n-2}), so entry t carries the exponent of
    // step t - 1: the gradient pass's row sums are those of 2^ah[t-1], i.e. 2^ex times the recursion's.
    __amdgpu_buffer_rsrc_t rsk = make_rsrc((R *) W.klog + (int64_t) b * T * 2, (!BETA && STORE && !O.no_store) ? (unsigned) T * 8u : 0u);
    auto frame = [&](int n) { return BETA ? len - 1 - n : n; };
    bool bad = false;
    // X (and block 0 of the rings) are there once the producer has published its first block
    {
        int spins = 0;
        while (lds_load_acq(&L.e_prod) < 1) {
            if (++spins > kSpinCap || lds_load_rlx(&L.kill)) { bad = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    const R XX = bad ? NINF : lds_ldf(&L.x[lane]);
    R sv = act ? Num<R>::exp2(-XX) : R(0);
    auto wait_slot = [&](int m) {            // until main has written s_m (main writes in order)
        float *slot = &L.s[m & (kRing - 1)][lane];
        int spins = 0;
        while (true) {
            const R v = lds_ldf(slot);
            if (__ballot(__float_as_uint(v) != kSentinel) == ~0ull) return true;
            if (++spins > kSpinCap || lds_load_rlx(&L.kill)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    if (STORE && !bad) buf_store((BETA ? XX : lds_ldf(&L.a[0][lane])) + Num<R>::log2(sv), rs, voff, (unsigned) frame(0) * row_bytes);
    if (STORE && !BETA && !bad) buf_store2(V2<R>{R(kScaleLogMark), R(0)}, rsk, lane == 0 ? 0u : kOobOffset, 0u);
    int n = 1;
    constexpr int GS = 8;                     // frames per poll: amortises the LDS round trips
    while (n < len && !bad) {
        // frames n .. n+g-1; a short last group re-processes its last frame in the unused positions (same values,
        // same addresses), which keeps the code free of predicates
        const int g = min(GS, len - n);
        if (!wait_slot(n + g - 2)) { bad = true; break; }
        if (!BETA) {      // arg (and the block scale) of the group's last frame: the producer is normally a block ahead of this
            int spins = 0;
            while (lds_load_rlx(&L.e_prod) < n + g) {
                if (++spins > kSpinCap || lds_load_rlx(&L.kill)) { bad = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (bad) break;
            asm volatile("" ::: "memory");
        }
        R sg[GS], ag[GS];
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            const int m = n + min(q, g - 1);
            sg[q] = lds_ldf(&L.s[(m - 1) & (kRing - 1)][lane]);
            ag[q] = BETA ? XX : lds_ldf(&L.a[m & (kRing - 1)][lane]);
        }
        unsigned lo = 0xffffffffu, hi = 0;
#pragma unroll
        for (int q = 0; q < GS; ++q) {
            const int m = n + min(q, g - 1);
            lds_stf(&L.s[(m - 1) & (kRing - 1)][lane], __uint_as_float(kSentinel));
            const unsigned sb = Rng<R>::bits(sg[q]);
            lo = min(lo, sb);
            hi = max(hi, sb);
        }
        // the values (incl. arg) are in registers: the producer may now refill (it waits for c_done)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_store_rlx(&L.c_done, n + g - 1);
        if ((__ballot(lo < Rng<R>::lo || hi > Rng<R>::hi) & actmask) != 0) { bad = true; break; }
        if (STORE) {
#pragma unroll
            for (int q = 0; q < GS; ++q)
                buf_store(ag[q] + Num<R>::log2(sg[q]), rs, voff, (unsigned) frame(n + min(q, g - 1)) * row_bytes);
        }
        if (STORE && !BETA) {
            // groups start at n = 1 (mod 8): frames n + 3 and n + 7 follow a rescaled step
            static_assert(GS == 8 && kRenorm == 4, "scale log of the three-wavefront chain");
            const int ex3 = __builtin_amdgcn_readlane(Rng<R>::expo(sg[1]), N), ex7 = __builtin_amdgcn_readlane(Rng<R>::expo(sg[5]), N);
            const R zbv = lds_ldf(&L.zb[((n + lane) >> 4) & 3]);
            const R kxv = lane == 3 ? (R) ex3 : (lane == 7 ? (R) ex7 : R(0));
            buf_store2(V2<R>{zbv, kxv}, rsk, lane < g ? (unsigned) (n + lane) * 8u : kOobOffset, 0u);
        }
        sv = sg[GS - 1];
        n += g;
    }
    if (!bad) {
        int spins = 0;
        while (!(lds_load_acq(&L.main_done) && lds_load_acq(&L.prod_done))) {
            if (++spins > kSpinCap || lds_load_rlx(&L.kill)) { bad = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (bad) {
#ifdef ASG_PROBE
        if (lane == 0) ((long long *) W.dbg)[16 + (BETA ? 1 : 0)] = 2;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // our stores land before main overwrites them
        lds_store_rel(&L.verdict, 2);
        return;
    }
    const int csum = lds_load_rlx(&L.csum);
    const double zsum = L.zsum;
    const R vlast = sv * lds_ldf(&L.e[(len - 1) & (kRing - 1)][lane]);
    const R sm = wave_allsum(act ? vlast : R(0));
    const double sc = zsum + (double) csum + (double) Num<R>::log2(sm);
    // same window as the row sums: a last emission factor far below its block's scale leaves a denormal sum, whose
    // v_log_f32 is -inf; the exact path then decides (as it does for all -inf, overflow, NaN)
    const unsigned smb = Rng<R>::bits(sm);
    const bool finite_ok = smb >= Rng<R>::lo && smb <= Rng<R>::hi;
#ifdef ASG_PROBE
    if (lane == 0) { ((long long *) W.dbg)[16 + (BETA ? 1 : 0)] = finite_ok ? 1 : 4; ((double *) W.dbg)[18 + (BETA ? 1 : 0)] = sc; ((double *) W.dbg)[22 + (BETA ? 1 : 0)] = zsum; ((long long *) W.dbg)[24 + (BETA ? 1 : 0)] = csum; }
#endif
    if (!finite_ok) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_store_rel(&L.verdict, 2);
        return;
    }
    lds_store_rlx(&L.verdict, 1);                      // main may leave: nothing to redo
    if (BETA) publish_score<R>(O, (R *) O.full_scores, b, P.B, score_out<R>(sc), lane);
    else if (O.full_scores_alpha && lane == 0) ((R *) O.full_scores_alpha)[b] = score_out<R>(sc);
}

template <int NP, bool STORE>
__global__ void __launch_bounds__(192, 1) fwd_duo_kernel(Problem P, State W, FwdOut O, int chain_mask) {
    __shared__ DuoLds L;
    int which = 0, seen = 0;
    for (int c = 0; c < 4; ++c) {
        if (chain_mask & (1 << c)) {
            if (seen == (int) blockIdx.y) which = 1 << c;
            ++seen;
        }
    }
    const int b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (which == kFullAlpha || which == kFullBeta) {
        if (threadIdx.x == 0) {
            L.e_prod = 0; L.verdict = 0; L.csum = 0; L.main_done = 0; L.kill = 0; L.c_done = 0; L.prod_done = 0;
        }
        for (int q = threadIdx.x; q < kRing * 64; q += 192) (&L.s[0][0])[q] = __uint_as_float(kSentinel);   // empty s ring
        __syncthreads();
        if (which == kFullAlpha) {
            if (wave == 0) duo_main<NP, STORE, false>(P, W, O, b, L);
            else if (wave == 1) duo_producer<NP, false>(P, b, L);
            else duo_consumer<NP, STORE, false>(P, W, O, b, L);
        } else {
            if (wave == 0) duo_main<NP, STORE, true>(P, W, O, b, L);
            else if (wave == 1) duo_producer<NP, true>(P, b, L);
            else duo_consumer<NP, STORE, true>(P, W, O, b, L);
        }
    } else if (wave == 0) {
        if (which == kAlignedAlpha) aligned_alpha_chain<float, STORE>(P, W, O, b);
        else if (which == kAlignedBeta) aligned_beta_chain<float, STORE>(P, W, O, b);
    }
}
}  // namespace
}  // namespace asg
