// NOT BUILT, NOT SHIPPED: the round-2 experiment (sixteen utterances per ONE wavefront), kept for the record with its numbers in
// README.md.  The product code of that idea's successor is torch_asg_amd/csrc/asg_batched.h + asg_batched.hip (round 4, a group of
// sixteen utterances spread over the wavefronts of a workgroup); the first line below names the path this file had when it was built.
//
// torch_asg_amd/csrc/asg_batched.h -- full-lattice alpha / beta recursions for LARGE batches (fp32, N <= 64):
// SIXTEEN utterances per wavefront, the per-step product  s[i][b] = sum_j E[i][j] u[j][b]  on the matrix cores.
//
// Replaces, for batches with many more chains than SIMDs, the one-utterance-per-wavefront chains of asg_chains.h
// (/root/reference/torch_asg/native/fully_connected_lattice.cpp:9-47, 65-91).  There every wavefront re-reads its
// utterance's vector through an LDS broadcast and spends 20 v_pk_fma_f32 + ~45 other issue slots per step, and the
// chip saturates on VALU issue at ~6 % of the HBM rate (batch sweep).  Here the transition matrix is the A
// operand of v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered), held in registers for the whole chain; the B operand is
// the state of sixteen utterances, and the accumulator layout IS the next step's B layout, so a step moves no data:
//   lane l = 16 g + n holds utterance b0 + n; register ks holds label 4 ks + g;
//   k-step ks of the product consumes register ks as it is; accumulator tile r', register i' = register ks = 4 r' + i'.
// Per step and utterance that leaves ~1/6 of the VALU work of the per-utterance chain (emission factor, one multiply,
// the log2 of the stored state) and N/4 * ceil(N/16) matrix instructions per sixteen utterances.
// Numerics are those of the per-utterance chains: exp domain, row-/column-normalised matrix, per-frame emission
// maximum and a power-of-two rescale by the L1 norm of the vector that enters the product, offsets summed in double;
// a norm outside [2^-100, 2^100] flags the utterance, which the per-utterance kernel then redoes (exact path included).
#pragma once
#include "asg_chains.h"
#include "asg_outer.h"

namespace asg {
namespace {

// sum / max over the four lane groups (lanes n, n + 16, n + 32, n + 48), result in all of them
__device__ __forceinline__ float grp_sum(float x) {
    float a = x, b = x;
    swap_halves(a, b);          // a = [x.lo | x.lo], b = [x.hi | x.hi]
    float s = a + b;
    float c = s, d = s;
    swap_rows(c, d);
    return c + d;
}
__device__ __forceinline__ float grp_max(float x) {
    float a = x, b = x;
    swap_halves(a, b);
    float s = fmaxf(a, b);
    float c = s, d = s;
    swap_rows(c, d);
    return fmaxf(c, d);
}

// Scores of up to sixteen utterances at once (lanes with `pub`): the group draws `cnt` tickets with one atomic.
__device__ __forceinline__ void publish_scores_x16(const FwdOut &O, float *slot, int b, int B, float score, bool pub, int lane) {
    if (!O.loss) {
        if (pub) slot[b] = score;
        return;
    }
    const unsigned long long pm = __ballot(pub);
    const unsigned cnt = (unsigned) __popcll(pm);
    if (cnt == 0) return;
    if (pub) __hip_atomic_store(slot + b, score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(O.counter, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket + cnt != (unsigned) O.expected) return;
    const float *full = (const float *) O.full_scores, *ali = (const float *) O.aligned_scores;
    float *loss = (float *) O.loss;
    double s = 0;
    for (int q = lane; q < B; q += 64) {
        float f = __hip_atomic_load(full + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float a = __hip_atomic_load(ali + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float l = f - a;
        if (O.reduction == 0) loss[q] = l;
        s += (double) l;
    }
    if (O.reduction != 0) {
        s = wave_allsum(s);
        if (lane == 0) loss[0] = (float) (O.reduction == 2 ? s / B : s);
    }
    if (lane == 0) __hip_atomic_store(O.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One direction of the full lattice for utterances 16 grp .. 16 grp + 15.  flags: [2][B] (alpha | beta), written for
// every utterance of the group: 1 = redo this direction with the per-utterance chain.
template <int NP, bool BETA, bool STORE>
__device__ void full_chain_x16(const Problem &P, const State &W, const FwdOut &O, int grp, int *flags, float *lds) {
    typedef float R;
    constexpr int KS = NP / 4, NT = (NP + 15) / 16;
    __builtin_amdgcn_s_setprio(3);      // sixteen chains ride on this wavefront; the aligned chains sharing the SIMD yield
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int N = P.N, T = P.T, B = P.B;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const int b = grp * 16 + n;
    const bool uv = b < B;
    const int bc = uv ? b : B - 1;
    const int len = uv ? (P.in_len ? clampi(P.in_len[bc], 0, T) : T) : 0;
    int tmax = len;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));
    tmax = __builtin_amdgcn_readfirstlane(tmax);

    // ---- normaliser of every row (alpha) / column (beta) of the transition matrix, one label per lane
    const R *tr = (const R *) P.transition;
    const int64_t so = BETA ? P.ts1 : P.ts0, si = BETA ? P.ts0 : P.ts1;      // stride of the output / input label
    {
        const int lc = lane < N ? lane : 0;
        R mx = NINF;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, tr[(int64_t) lc * so + (int64_t) j * si] * L2E);
        if (!(mx > NINF)) mx = 0;           // -inf (or NaN) line: as load_norm_row
        lds[lane] = mx;
        __builtin_amdgcn_wave_barrier();
        if (!BETA && STORE && grp == 0 && lane < N) {
            // publish the normalised rows once per forward: the gradient assembly reads them
            R *erow = (R *) W.ehat + (int64_t) lane * W.npad;
            for (int j = 0; j < W.npad; ++j)
                erow[j] = j < N ? Num<R>::exp2(tr[(int64_t) lane * so + (int64_t) j * si] * L2E - mx) : R(0);
            ((R *) W.rmax)[lane] = mx;
        }
    }
    // ---- A operand: tile r', k-step ks; lane (k-slot g, row m = n) -> E[out label 16 r' + 4 (m & 3) + (m >> 2)][in label 4 ks + g]
    R A[NT][KS];
#pragma unroll
    for (int r = 0; r < NT; ++r) {
        const int o = 16 * r + 4 * (n & 3) + (n >> 2);
        const int oc = o < N ? o : 0;       // padding rows repeat label 0's row: their sums are ordinary numbers (see watch)
        const R mo = lds[oc];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int in = 4 * ks + g, ic = in < N ? in : 0;
            const R v = Num<R>::exp2(tr[(int64_t) oc * so + (int64_t) ic * si] * L2E - mo);
            A[r][ks] = in < N ? v : R(0);
        }
    }
    // ---- per register: normaliser of its label (-inf on padding: its emission factor is then 0), load / store offsets
    R M[KS];
    unsigned xoff[KS], soff[KS];
    const unsigned kBad = kOobOffset;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int lab = 4 * ks + g;
        const bool lv = lab < N;
        M[ks] = lv ? lds[lv ? lab : 0] : NINF;
        xoff[ks] = (unsigned) ((int64_t) bc * P.is1 + (int64_t) (lv ? lab : 0) * P.is2) * 4u;
        soff[ks] = (lv && uv) ? (unsigned) (((int64_t) b * T) * N + lab) * 4u : kBad;
    }
    __amdgpu_buffer_rsrc_t rin = make_rsrc((R *) P.inputs, 0xffffffffu);
    __amdgpu_buffer_rsrc_t rst = make_rsrc(BETA ? (R *) W.bh : (R *) W.ah, STORE ? (unsigned) ((int64_t) B * T * N) * 4u : 0u);
    const unsigned fstride = (unsigned) P.is0 * 4u, row_bytes = (unsigned) N * 4u;

    auto load_frame = [&](R (&x)[KS], int t) {
        const unsigned so2 = (unsigned) clampi(t, 0, T - 1) * fstride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x[ks] = buf_load<R>(rin, xoff[ks], so2);
    };
    // tl = x log2e + M (padding: -inf), m = its maximum over the utterance's labels
    auto frame_max = [&](const R (&x)[KS], R (&tl)[KS]) -> R {
        R mx = NINF;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            tl[ks] = fmaf(x[ks], L2E, M[ks]);
            mx = fmaxf(mx, tl[ks]);
        }
        return fmaxf(grp_max(mx), LZ);
    };
    auto product = [&](const R (&u)[KS], V4<R> (&d)[NT]) {
#pragma unroll
        for (int r = 0; r < NT; ++r) d[r] = V4<R>{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int r = 0; r < NT; ++r) d[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r][ks], u[ks], d[r], 0, 0, 0);
    };
    auto norm_of = [&](const R (&u)[KS]) -> R {
        R s = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s += u[ks];
        return grp_sum(s);
    };

    bool flagged = false;
    double C = 0.0;
    R u[KS], x[KS], xn[KS], tl[KS];
    // sticky extremes of the row sums' bit patterns (as the per-utterance chain: a row sum outside 2^+-100 -- zero
    // included -- sends the utterance to the exact code); padding rows count as 1.0
    unsigned wlo = 0xffffffffu, whi = 0u;
    auto watch = [&](const V4<R> (&d)[NT], bool act) {
        unsigned lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned sb = Rng<R>::bits(d[ks >> 2][ks & 3]);
            lo = min(lo, sb);
            hi = max(hi, sb);
        }
        wlo = act ? min(wlo, lo) : wlo;
        whi = act ? max(whi, hi) : whi;
    };

    // One step in program order: [matrix product of the current vector] [everything that does not depend on it: the
    // previous state's log2 and store, the norm / rescale exponent, the next frame's emission factors] [row-sum watch,
    // one multiply].  The sched_group_barrier pipeline asks hipcc to interleave the middle part with the 30 MFMAs (a
    // lone wavefront issues in order: MFMAs issued back to back would leave the VALU idle for ~900 cycles per step).
#define ASG_X16_PIPELINE()                                                \
    _Pragma("unroll") for (int q_ = 0; q_ < NT * KS; ++q_) {              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                \
        __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);                \
    }

    if (!BETA) {
        // ---------------------------------------------------------------- alpha: u_0 = exp2(I2_0 - max), then t = 1 ..
        R scoreA = NINF;
        load_frame(x, 0);
        load_frame(xn, 1);
        {
            R mx = NINF;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                tl[ks] = M[ks] > NINF ? x[ks] * L2E : NINF;
                mx = fmaxf(mx, tl[ks]);
            }
            mx = fmaxf(grp_max(mx), LZ);
            C = (double) mx;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const R a0 = tl[ks] - mx;
                u[ks] = Num<R>::exp2(a0);
                if (STORE) buf_store(a0, rst, len >= 1 ? soff[ks] : kBad, 0u);
            }
        }
        for (int t = 1; t < tmax; ++t) {
            const bool act = t < len;
            V4<R> d[NT];
            product(u, d);
            // ---- independent of the product: u_{t-1} -> stored state, norm; frame t -> emission factors
            if (STORE) {
                const unsigned pmask = (t - 1 >= 1 && t - 1 < len) ? 0u : kBad;      // row 0 is already stored exactly
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    buf_store(Num<R>::log2(u[ks]), rst, max(soff[ks], pmask), (unsigned) (t - 1) * row_bytes);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) x[ks] = xn[ks];
            load_frame(xn, t + 1);
            const R nrm = norm_of(u);
            scoreA = t == len ? score_out<R>(C + (double) Num<R>::log2(nrm)) : scoreA;
            const R c = frame_max(x, tl) + (R) Rng<R>::expo(nrm);
            C += act ? (double) c : 0.0;
            R e[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) e[ks] = Num<R>::exp2(tl[ks] - c);
            // ---- dependent on the product
            watch(d, act);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) u[ks] = d[ks >> 2][ks & 3] * e[ks];
            ASG_X16_PIPELINE()
        }
        if (STORE && tmax >= 2) {
            const unsigned pmask = (tmax - 1 < len) ? 0u : kBad;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                buf_store(Num<R>::log2(u[ks]), rst, max(soff[ks], pmask), (unsigned) (tmax - 1) * row_bytes);
        }
        if (O.full_scores_alpha) {
            const R nrm = norm_of(u);
            if (len == tmax && len >= 1) scoreA = score_out<R>(C + (double) Num<R>::log2(nrm));
            if (g == 0 && uv) ((R *) O.full_scores_alpha)[b] = scoreA;
        }
        flagged = wlo < Rng<R>::lo || whi > Rng<R>::hi;
        if (g == 0 && uv) flags[b] = flagged ? 1 : 0;
        return;
    }

    // -------------------------------------------------------------------- beta: joins at frame len - 1 with beta = 0
    R s[KS], sinit[KS], e[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        sinit[ks] = M[ks] > NINF ? Num<R>::exp2(-M[ks]) : R(0);
        s[ks] = 0;
    }
    if (STORE && len >= 1) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) buf_store(R(0), rst, soff[ks], (unsigned) (len - 1) * row_bytes);
    }
    load_frame(x, tmax - 1);
    load_frame(xn, tmax - 2);
    R c = frame_max(x, tl);          // frame tmax - 1: whoever is consumed there joins there (k = 0)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) e[ks] = Num<R>::exp2(tl[ks] - c);
    for (int t = tmax - 1; t >= 1; --t) {
        // frame t: u = s o e_t, then s <- F u  (= beta_{t-1})
        const bool act = t <= len - 1, join = t == len - 1;
        C += act ? (double) c : 0.0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) u[ks] = act ? (join ? sinit[ks] : s[ks]) * e[ks] : R(0);
        V4<R> d[NT];
        product(u, d);
        // ---- independent of the product: beta_t -> stored state, norm, frame t - 1 -> emission factors
        if (STORE) {
            const unsigned pmask = (act && !join) ? 0u : kBad;        // row len - 1 holds its exact zeros already
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                buf_store(M[ks] + Num<R>::log2(s[ks]), rst, max(soff[ks], pmask), (unsigned) t * row_bytes);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x[ks] = xn[ks];
        load_frame(xn, t - 2);
        const R nrm = norm_of(u);
        const bool join_n = t - 1 == len - 1;
        c = frame_max(x, tl) + (join_n ? R(0) : (R) Rng<R>::expo(nrm));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) e[ks] = Num<R>::exp2(tl[ks] - c);
        // ---- dependent on the product
        watch(d, act);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[ks] = d[ks >> 2][ks & 3];
        ASG_X16_PIPELINE()
    }
    if (STORE) {
        const unsigned pmask = len >= 2 ? 0u : kBad;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) buf_store(M[ks] + Num<R>::log2(s[ks]), rst, max(soff[ks], pmask), 0u);
    }
    // frame 0: S_full = LSE_i(I2_0[i] + beta_0[i])
    {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            if (len == 1) s[ks] = sinit[ks];
        const R m0 = frame_max(x, tl);      // x: frame 0 (the loop's last look-ahead, or the first load when tmax <= 1)
        R sm = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) sm += s[ks] * Num<R>::exp2(tl[ks] - m0);
        sm = grp_sum(sm);
        // a vanished or overflowed total is the per-utterance chain's business too
        flagged = wlo < Rng<R>::lo || whi > Rng<R>::hi || (len >= 1 && !(sm > 0 && sm < __builtin_inff()));
        const R score = len >= 1 ? score_out<R>(C + (double) m0 + (double) Num<R>::log2(sm)) : NINF;
        if (g == 0 && uv) flags[B + b] = flagged ? 1 : 0;
        publish_scores_x16(O, (R *) O.full_scores, b, B, score, g == 0 && uv && !flagged, lane);
    }
}

}  // namespace
}  // namespace asg
