// torch_asg_amd/csrc/asg_batched.h -- full-lattice alpha / beta recursions for LARGE batches (fp32, N <= 64):
// SIXTEEN utterances per workgroup, the per-step product  s[i][b] = sum_j E[i][j] u[j][b]  on the matrix cores, the
// label axis split over the wavefronts of the workgroup (one 16-label tile each).
//
// Replaces, for batches with many more chains than SIMDs, the one-utterance-per-wavefront chains of asg_chains.h
// (/root/reference/torch_asg/native/fully_connected_lattice.cpp:9-47, 65-91).  There every wavefront re-reads its
// utterance's vector through an LDS broadcast and spends 20 v_pk_fma_f32 + ~45 other issue slots per step: the chip
// saturates on VALU issue at ~6 % of the HBM rate (batch sweep).  Here the transition matrix is the A operand
// of v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered), held in registers for the whole chain; the B operand is the state of
// sixteen utterances.  Wavefront w of a workgroup owns output labels 16 w .. 16 w + 15:
//   lane l = 16 g + n holds utterance b0 + n and labels 16 w + 4 g + r, r = 0..3  ("natural": four consecutive labels, one
//   16-byte access per frame for emissions and states); the product's accumulator has exactly this layout;
//   the B operand wants label 4 q + g in register q -- a 4 x 4 transpose of (lane group, register), two rounds of
//   v_permlane32_swap / v_permlane16_swap -- after which the tile goes to the other wavefronts through LDS (one
//   ds_write_b128, NT - 1 ds_read_b128, ONE workgroup barrier per step, double-buffered).
// (Round 2 tried ONE wavefront per sixteen utterances -- tools/experiments/asg_batched.h: no data movement at all, and slower
// than the per-utterance chains up to B ~ 8192, because B = 4096 is only 512 such wavefronts for 1024 SIMDs and each
// serialises 30 MFMAs + 350 VALU instructions per step.  Splitting the label axis gives 3 x as many wavefronts a third
// of the work each; the alpha and beta workgroups of a compute unit fill each other's exchange latency.)
// Numerics are those of the per-utterance chains: exp domain, row- / column-normalised matrix, the block's emission
// maximum (16 frames) and a power-of-two rescale by the L1 norm of the vector that enters the product, offsets summed in
// double; a row sum outside [2^-100, 2^100] flags the utterance, which the per-utterance kernel then redoes (exact path
// included): launch_fwd_small's clean-up launch.
#pragma once
#include "asg_chains.h"
#include "asg_outer.h"

namespace asg {
namespace {

#ifndef ASG_X16_ABL
#define ASG_X16_ABL 0          // developer timing probes (wrong results): 1 no state stores, 2 no exchange, 4 no matrix instructions,
#endif                         // 8 no emission factors, 16 exchange without the barrier, 32 no log2 of the stored state
constexpr int kXB = 8;           // frames per emission block (one common scale per block and utterance)

struct X16Lds {
    __attribute__((aligned(16))) float xch[2][4][64][4];      // [parity][tile][lane][B register]: the step's vector
    float part[2][4][16];                                      // [parity][tile][utterance]: partial maxima (block scale)
    float mxs[4][64];                                          // per wavefront: row / column normalisers, one label per lane
    int flag[16];                                              // per utterance: a row sum left the safe range
};

// Workgroup barrier that waits for this wavefront's LDS operations only: __syncthreads() also drains the vector-memory queue
// (vmcnt(0)), i.e. every step would wait for the block of emission loads in flight and for the previous frame's state store.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Make values opaque at this point of the program: what computes them stays above (MachineSink moves a computation whose
// result is only used behind a later branch down into that block -- out of the matrix instructions' shadow), no consumer above.
__device__ __forceinline__ void pin(V4<float> &v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); }
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }

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

// One direction of the full lattice for utterances 16 grp .. 16 grp + 15, by the NT = ceil(NP / 16) wavefronts of the
// calling workgroup (all of them must call: the body contains workgroup barriers).
// flags: [2][B] (alpha | beta), written for every utterance of the group: 1 = redo this direction with the per-utterance chain.
// VEC: N % 4 == 0, unit label stride, 16-byte aligned rows: emissions and states move as one 16-byte access per frame and lane.
//
// Both directions are the same loop (X = row maximum R_i for alpha / column maximum C_i for beta, frames f_0, f_1, ...
// = 0, 1, ... for alpha and tmax-1, tmax-2, ... for beta; M = the normalised matrix):
//   vector_n = d_n o e_n ,   d_n = M vector_{n-1} ,   e_n = 2^(I2[f_n] + X - c_n) ,   c_n = block scale + a power-of-two exponent
//   alpha stores log2 vector_n at row f_n;  beta stores X + log2 d_n at row f_n  (= beta_hat of frame f_n)
// The exponent in c_n only has to keep the vector in range -- whatever is applied is summed (exactly: integers) into the
// utterance's offset -- so it is taken from the L1 norm that is cheapest to have: with padding rows in the last tile
// (N % 16 != 0, VEC) those rows of M are rows of ONES, the product itself returns |vector_{n-1}|_1 there, it rides to
// every wavefront with the exchanged tile, and c_{n+1} uses it (one step late); otherwise the norm of vector_{n-1} is summed
// on the VALU from the B operand.
// An utterance shorter than the group's longest JOINS the beta loop at its own last frame (vector = 2^-X o e there; before
// that it runs the same recursion on its clamped last frame: ordinary numbers that are overwritten at the join) and simply
// keeps running past its end in the alpha loop; stores, offsets and scores are masked per lane.
template <int NP, bool BETA, int MODE>
__device__ void full_chain_x16(const Problem &P, const State &W, const FwdOut &O, int grp, int *flags, X16Lds &L) {
    typedef float R;
    constexpr int KS = NP / 4, NT = (NP + 15) / 16;
    // sixteen chains ride on every wavefront of this kernel, and each step is a dependent chain of instructions: the aligned
    // chains that share the SIMD (one utterance or two per wavefront) yield to it and take the issue slots it leaves
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    if (lane < 16 && w == 0) L.flag[lane] = 0;
    const bool want_score = BETA || O.full_scores_alpha != nullptr;
    // MODE 0: any strides / N (scalar accesses, norm on the VALU); 1: VEC; 2: VEC and the L1 norm from the product's padding
    // rows, which needs a whole k-step of padding labels: N % 4 == 0 and N % 16 != 0 (the launcher checks)
    constexpr bool VEC = MODE >= 1, ones = MODE == 2;

    // ---- normaliser of every row (alpha) / column (beta) of the transition matrix, one label per lane (every wavefront its own copy)
    const R *tr = (const R *) P.transition;
    const int64_t so = BETA ? P.ts1 : P.ts0, si = BETA ? P.ts0 : P.ts1;      // stride of the output / input label
    {
        const int lc = lane < N ? lane : 0;
        R mx = NINF;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, tr[(int64_t) lc * so + (int64_t) j * si] * L2E);
        if (!(mx > NINF)) mx = 0;           // -inf (or NaN) line: as load_norm_row
        L.mxs[w][lane] = mx;
        __builtin_amdgcn_wave_barrier();
        if (!BETA && !O.no_store && grp == 0 && w == 0 && lane < N) {
            // publish the normalised rows once per forward: the gradient assembly reads them
            R *erow = (R *) W.ehat + (int64_t) lane * W.npad;
            for (int j = 0; j < W.npad; ++j)
                erow[j] = j < N ? Num<R>::exp2(tr[(int64_t) lane * so + (int64_t) j * si] * L2E - mx) : R(0);
            ((R *) W.rmax)[lane] = mx;
        }
    }
    // ---- A operand of k-step q: lane (k-slot g, row m = n) -> M[out label 16 w + m][in label 4 q + g]; padding rows: ones
    R A[KS];
    {
        const int o = 16 * w + n;
        const int oc = o < N ? o : 0;
        const R mo = L.mxs[w][oc];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int in = 4 * q + g, ic = in < N ? in : 0;
            const R v = Num<R>::exp2(tr[(int64_t) oc * so + (int64_t) ic * si] * L2E - mo);
            A[q] = in < N ? (o < N ? v : R(1)) : R(0);
        }
    }
    // ---- this lane's four labels 16 w + 4 g + r: normaliser, offsets.  A padding label's emission factor is 0 (X = -inf) --
    // or, in `ones` mode, exactly 1 (scale and offset 0), so that the norm in its slot travels on unchanged
    R X[4], sinit[4], L2Er[4];
    bool lv[4];
    unsigned xoff[4], soff[4];
    const int lab0 = 16 * w + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lab = lab0 + r;
        lv[r] = lab < N;
        X[r] = lv[r] ? L.mxs[w][lv[r] ? lab : 0] : (ones ? R(0) : NINF);
        L2Er[r] = (lv[r] || !ones) ? L2E : R(0);
        sinit[r] = lv[r] ? Num<R>::exp2(-X[r]) : R(0);
        xoff[r] = (unsigned) ((int64_t) bc * P.is1 + (int64_t) (lv[r] ? lab : 0) * P.is2) * 4u;
        soff[r] = (lv[r] && uv) ? (unsigned) (((int64_t) b * T) * N + lab) * 4u : kOobOffset;
    }
    const bool padlane = !lv[0];                   // (VEC: a lane's four labels are valid or padding together)
    // `ones` mode: a padding lane's emission must read as an ordinary number (its factor is 2^(0 * x + 0) = 1, and 0 * -inf is
    // NaN): it loads from beyond the tensor's last byte -- the buffer resource ends there, such loads return 0
    const unsigned in_bytes = ones ? (unsigned) (((int64_t) (T - 1) * P.is0 + (int64_t) (B - 1) * P.is1 + (int64_t) (N - 1) * P.is2 + 1) * 4)
                                   : 0xffffffffu;
    if (ones && padlane) xoff[0] = 0xfffffff0u;
    __amdgpu_buffer_rsrc_t rin = make_rsrc((R *) P.inputs, in_bytes);
    __amdgpu_buffer_rsrc_t rst = make_rsrc(BETA ? (R *) W.bh : (R *) W.ah, O.no_store ? 0u : (unsigned) ((int64_t) B * T * N) * 4u);
    const unsigned fstride = (ones && padlane) ? 0u : (unsigned) P.is0 * 4u, row_bytes = (unsigned) N * 4u;
    const int lenm1 = len >= 1 ? len - 1 : 0;
    // alpha: the scale log of the stored states (ScaleLog, asg_kernels.h), one entry per frame and utterance, by lane group 0 of
    // wavefront 0
    __amdgpu_buffer_rsrc_t rsk = make_rsrc((R *) W.klog, (!BETA && !O.no_store) ? (unsigned) ((int64_t) B * T * 2) * 4u : 0u);
    const bool klane = !BETA && w == 0 && g == 0 && uv;
    const unsigned koff = (unsigned) ((int64_t) b * T * 2) * 4u;

    // frame `t` of this lane's utterance (clamped to the utterance: what lies beyond its end is never read)
    auto load_frame = [&](V4<R> &x, int t) {
        const unsigned fo = (unsigned) min(max(t, 0), lenm1) * fstride;
        if constexpr (VEC) {
            x = __builtin_bit_cast(V4<R>, __builtin_amdgcn_raw_buffer_load_b128(rin, xoff[0] + fo, 0u, 0));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = buf_load<R>(rin, xoff[r] + fo, 0u);
        }
    };
    auto store_row = [&](const V4<R> &v, int t, bool ok) {
        if (ASG_X16_ABL & 1) return;
        const unsigned ro = (unsigned) t * row_bytes;
        if constexpr (VEC) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rst,
                                                   ok ? soff[0] + ro : kOobOffset, 0u, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) buf_store(v[r], rst, ok ? soff[r] + ro : kOobOffset, 0u);      // (invalid: >= 2^31, past any buffer)
        }
    };
    // frame index of step n
    auto frame_of = [&](int nn) { return BETA ? tmax - 1 - nn : nn; };

    // the vector of sixteen utterances, all labels, in B-operand order: register q, lane (g, n) = label 4 q + g
    R vb[4 * NT];
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) vb[q] = 0;
    // own tile (natural layout) -> B-operand order -> the other wavefronts; one barrier; everybody's tile back
#ifdef ASG_X16_PROBE
    long long prb[6] = {0, 0, 0, 0, 0, 0}, pq0 = 0, pq1 = 0;      // cycles: transposes + write, barrier, read, products + shadow, tail, steps
#define ASG_XPRB(stmt) stmt
#else
#define ASG_XPRB(stmt)
#endif
    auto exchange = [&](const V4<R> &mine, int par) {
        ASG_XPRB(const long long pa = clock64(); prb[4] += pa - pq1;)
        R tb[4] = {mine[0], mine[1], mine[2], mine[3]};
        frames_to_operands(tb);
        if (ASG_X16_ABL & 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) vb[(4 * w + q) % (4 * NT)] = tb[q];
            return;
        }
        if (NT > 1) {
            *reinterpret_cast<V4<R> *>(&L.xch[par][w][lane][0]) = V4<R>{tb[0], tb[1], tb[2], tb[3]};
            ASG_XPRB(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long pb = clock64(); prb[0] += pb - pa;)
            if (ASG_X16_ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else wg_barrier();
            ASG_XPRB(const long long pc = clock64(); prb[1] += pc - pb;)
#pragma unroll
            for (int ww = 0; ww < NT; ++ww) {
                const V4<R> o = *reinterpret_cast<const V4<R> *>(&L.xch[par][ww][lane][0]);
#pragma unroll
                for (int q = 0; q < 4; ++q) vb[4 * ww + q] = o[q];
            }
            ASG_XPRB(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); prb[2] += clock64() - pc;)
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) vb[q] = tb[q];
        }
    };
    // L1 norm of the exchanged vector, summed on the VALU (padding slots hold 0 -- or, in `ones` mode, lie outside the sum:
    // the first padding k-step is ceil(N / 4) >= the number of k-steps that carry labels)
    auto norm_of = [&]() -> R {
        R s0 = 0, s1 = 0;
#pragma unroll
        for (int q = 0; q < KS; q += 2) {
            s0 += (ones && 4 * q >= N) ? R(0) : vb[q];
            if (q + 1 < KS) s1 += (ones && 4 * (q + 1) >= N) ? R(0) : vb[q + 1];
        }
        return grp_sum(s0 + s1);
    };
    // `ones` mode: the norm the LAST product found, in the first padding k-step of the exchanged vector (all lane groups)
    auto norm_slot = [&]() -> R {
        if constexpr (KS < 4 * NT) return (N <= NP - 4) ? vb[KS - 1] : vb[KS];
        else return vb[KS - 1];
    };
    // per-utterance maximum over this wavefront's labels of x log2 e + X, then over the wavefronts (partial maxima through LDS)
    auto tile_max = [&](const V4<R> &zl) -> R {
        R mx = NINF;
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, lv[r] ? fmaf(zl[r], L2E, X[r]) : NINF);
        return grp_max(mx);
    };
    auto all_tiles_max = [&](int par) -> R {
        R m = L.part[par][0][n];
#pragma unroll
        for (int ww = 1; ww < NT; ++ww) m = fmaxf(m, L.part[par][ww][n]);
        return fmaxf(m, LZ);
    };

    // sticky extremes of the row sums' bit patterns (as the per-utterance chain: a row sum outside 2^+-100 -- zero
    // included -- sends the utterance to the exact code).  Padding rows return the vector's norm: an ordinary number.
    unsigned wlo = 0xffffffffu, whi = 0u;
    auto watch = [&](const V4<R> &d) {
        const unsigned b0 = Rng<R>::bits(d[0]), b1 = Rng<R>::bits(d[1]), b2 = Rng<R>::bits(d[2]), b3 = Rng<R>::bits(d[3]);
        wlo = min(min(wlo, b0), min(b1, min(b2, b3)));
        whi = max(max(whi, b0), max(b1, max(b2, b3)));
    };

    // ---- emission blocks: block k = steps kXB k .. kXB k + 15; cur = the running block, nxt = the next one (in flight)
    V4<R> cur[kXB], nxt[kXB];
    const int nsteps = tmax;                       // steps n = 0 .. tmax - 1 (step n handles frame f_n)
    auto load_block = [&](V4<R> (&dst)[kXB], int k) {
#pragma unroll
        for (int f = 0; f < kXB; ++f) load_frame(dst[f], frame_of(kXB * k + f));
    };
    // partial block scale of `blk` into LDS slot `par` (read back after the next barrier)
    auto post_block_max = [&](const V4<R> (&blk)[kXB], int k, int par) {
        V4<R> zl = blk[0];
#pragma unroll
        for (int f = 1; f < kXB; ++f) {
            const bool in = kXB * k + f < nsteps;          // (frames past the group's last step: clamped repeats, harmless)
#pragma unroll
            for (int r = 0; r < 4; ++r) zl[r] = in ? fmaxf(zl[r], blk[f][r]) : zl[r];
        }
        const R pm = tile_max(zl);
        if (g == 0) L.part[par][w][n] = pm;
    };

    if (tmax < 1) {           // nothing to do for the whole group
        if (w == 0) {
            if (g == 0 && uv) flags[(BETA ? B : 0) + b] = 0;
            if (BETA) publish_scores_x16(O, (R *) O.full_scores, b, B, NINF, g == 0 && uv, lane);
            else if (O.full_scores_alpha && g == 0 && uv) ((R *) O.full_scores_alpha)[b] = NINF;
        }
        return;
    }
    load_block(cur, 0);
    load_block(nxt, 1);
    post_block_max(cur, 0, 0);
    wg_barrier();
    R zb = all_tiles_max(0);
    R Xz[4];                          // X - block scale (padding labels in `ones` mode: 0, their factor stays 1)
    auto set_scale = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) Xz[r] = X[r] - ((lv[r] || !ones) ? zb : R(0));
    };
    set_scale();

    // The utterance's offset: C = Cd + csum, Cd = the block scales of its active steps (summed per block, in double), csum = the
    // power-of-two exponents applied at its active steps (integers).  First active step: alpha 0, beta nj = tmax - len.
    const int nj = BETA ? tmax - len : 0;
    double Cd = 0.0;
    int csum = 0;
    R score = NINF;
    V4<R> mine = {0, 0, 0, 0};       // own tile of the current vector (natural layout)
    // the state row of the previous step is written inside the next product's shadow: `keep` is the linear value (alpha: the
    // vector itself, beta: the product) whose log2 is taken there
    V4<R> keep = {1, 1, 1, 1};
    bool keep_ok = false;
    int keep_t = 0;
    auto flush_keep = [&]() {
        V4<R> o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (BETA ? X[r] : R(0)) + Num<R>::log2(keep[r]);
        store_row(o, keep_t, keep_ok);
    };
    // active steps of block k (steps kXB k .. kXB k + 15, below nsteps) of this lane's utterance
    auto active_in_block = [&](int k) -> int {
        const int lo = max(kXB * k, nj), hi = BETA ? min(kXB * k + kXB, nsteps) : min(kXB * k + kXB, len);
        return max(hi - lo, 0);
    };

    // ---- step 0: the first vector
    {
        const int f0 = frame_of(0);
        if (!BETA) {
            // alpha_0 = I_0: no transition, no normaliser; common scale = the frame's maximum (instead of the block's: Cd takes it)
            V4<R> tl, row;
            R mx = NINF;
#pragma unroll
            for (int r = 0; r < 4; ++r) { tl[r] = lv[r] ? cur[0][r] * L2E : NINF; mx = fmaxf(mx, tl[r]); }
            mx = grp_max(mx);
            if (g == 0) L.part[1][w][n] = mx;
            wg_barrier();
            const R m0 = all_tiles_max(1);
            Cd = (double) m0 - (double) zb;      // (block 0 counts zb for step 0 as for any other active step)
#pragma unroll
            for (int r = 0; r < 4; ++r) { row[r] = tl[r] - m0; mine[r] = Num<R>::exp2(row[r]); }
            store_row(row, 0, len >= 1);
            buf_store2(V2<R>{R(kScaleLogMark), R(0)}, rsk, (klane && len >= 1) ? koff : kOobOffset, 0u);
        } else {
            // beta: the utterances whose last frame is f0 join here with beta = 0, i.e. d = 2^-X; the others run on their clamped
            // last frame until their own join overwrites them
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[r] = sinit[r] * Num<R>::exp2(fmaf(cur[0][r], L2Er[r], Xz[r]));
            // row len - 1 of beta_hat is exactly X + log2(2^-X) = 0
            store_row(V4<R>{0, 0, 0, 0}, f0, f0 == len - 1);
        }
        exchange(mine, 0);
    }

    // ---- steps 1 .. nsteps - 1, a block at a time (the block's frames are compile-time register indices).
    // One step in program order: the product's KS matrix instructions in five groups, and BETWEEN the groups everything that does
    // not depend on this product (the previous frame's log2 + store, the rescale exponent, this frame's emission factors): a
    // wavefront issues in order, so matrix instructions issued back to back leave its VALU idle for 32 cycles apiece, and
    // whatever follows them in program order starts after the last one.  sched_barrier pins the interleave (left alone, hipcc
    // put six of the ten products in a row and the stores behind them: 950 cycles per step).
    auto mfma_range = [&](V4<R> &d0, V4<R> &d1, int q0, int q1) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            if (q >= q0 && q < q1) {
                if (ASG_X16_ABL & 4) { d0[q & 3] += A[q] * vb[q]; continue; }
                if (q & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q], vb[q], d1, 0, 0, 0);
                else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[q], vb[q], d0, 0, 0, 0);
            }
        }
    };
    auto step = [&](int nn, const V4<R> &xr, bool post, int k) {
        constexpr int G = (KS + 4) / 5;               // matrix instructions per group
        const int ft = frame_of(nn);
        V4<R> d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
        ASG_XPRB(pq0 = clock64(); prb[5] += 1;)
        mfma_range(d0, d1, 0, G);
        __builtin_amdgcn_sched_barrier(0);
        // ---- (1) the previous frame's state: log2
        V4<R> o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (ASG_X16_ABL & 32) ? keep[r] : Num<R>::log2(keep[r]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(d0, d1, G, 2 * G);
        __builtin_amdgcn_sched_barrier(0);
        // ---- (2) ... and its store; the rescale exponent
        if (BETA) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += X[r];
        }
        store_row(o, keep_t, keep_ok);
        R nrm;
        if constexpr (ones) nrm = norm_slot();
        else nrm = norm_of();
        if (!BETA && want_score) {
            // (the norm of vector_m is needed where m = len - 1: `ones` mode sees it two steps later, else one)
            const int at = ones ? len + 1 : len;
            if (__any(nn == at)) score = (nn == at) ? score_out<R>(Cd + (double) csum + (double) Num<R>::log2(nrm)) : score;
        }
        const int ex = Rng<R>::expo(nrm);
        const R exf = (VEC && padlane && ones) ? R(0) : (R) ex;
        if (!BETA) buf_store2(V2<R>{zb, (R) ex}, rsk, (klane && nn < len) ? koff + (unsigned) nn * 8u : kOobOffset, 0u);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(d0, d1, 2 * G, 3 * G);
        __builtin_amdgcn_sched_barrier(0);
        // ---- (3) this frame's emission factors: arguments
        V4<R> e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = fmaf(xr[r], L2Er[r], Xz[r]) - exf;      // (rounded as the scale log says)
        if (BETA) csum += ex;
        else if (want_score) csum += nn < len ? ex : 0;
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(d0, d1, 3 * G, 4 * G);
        __builtin_amdgcn_sched_barrier(0);
        // ---- (4) ... and the exponentials; once per block the next block's scale
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = (ASG_X16_ABL & 8) ? e[r] : Num<R>::exp2(e[r]);
        if (post) post_block_max(nxt, k + 1, (k + 1) & 1);      // (read after this step's barrier, at the block switch)
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(d0, d1, 4 * G, KS);
        __builtin_amdgcn_sched_barrier(0);
        ASG_XPRB(pq1 = clock64(); prb[3] += pq1 - pq0;)
        // ---- dependent on the product
        const V4<R> d = d0 + d1;
        watch(d);
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = d[r] * e[r];
        keep = BETA ? d : mine;
        keep_ok = BETA ? nn > nj : nn < len;
        keep_t = ft;
        if (BETA) {
            if (__any(nn == nj)) {
                // joiners: vector = 2^-X o 2^(I2 + X - zb), offset restarted, row len - 1 of beta_hat is exactly 0
                const bool join = nn == nj;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mine[r] = join ? sinit[r] * Num<R>::exp2(fmaf(xr[r], L2Er[r], Xz[r])) : mine[r];
                csum = join ? 0 : csum;
                store_row(V4<R>{0, 0, 0, 0}, ft, join);
            }
        }
        exchange(mine, nn & 1);
    };
    for (int k = 0; kXB * k < nsteps; ++k) {
        if (k > 0) {
            // next block becomes current (its loads were issued a whole block ago), the one after it starts to travel
#pragma unroll
            for (int f = 0; f < kXB; ++f) cur[f] = nxt[f];
            load_block(nxt, k + 1);
            zb = all_tiles_max(k & 1);             // posted during the previous block's last step, before its barrier
            set_scale();
        }
        if (want_score) Cd += (double) zb * (double) active_in_block(k);
#pragma unroll
        for (int f = 0; f < kXB; ++f) {
            const int nn = kXB * k + f;
            if (nn >= 1 && nn < nsteps) step(nn, cur[f], f == kXB - 1, k);
        }
    }
#ifdef ASG_X16_PROBE
    if (blockIdx.x == 0 && w == 0 && lane == 0) {
        long long *dd = (long long *) W.dbg;
        dd[0] = 0x1234567890abcdefLL;
        for (int i = 0; i < 6; ++i) dd[1 + i] = prb[i];
    }
#endif
    // ---- the last vector: its state row, the score (the L1 norm of the vector that was exchanged last: for beta that is
    // sum_i 2^(I2_0[i] + beta_0[i]), S_full itself)
    {
        flush_keep();
        if (want_score) {
            const R nrm = norm_of();
            if (!BETA) {
                if (len == tmax && len >= 1) score = score_out<R>(Cd + (double) csum + (double) Num<R>::log2(nrm));
                if (ones && len == tmax - 1 && len >= 1) score = score_out<R>(Cd + (double) csum + (double) Num<R>::log2(norm_slot()));
            } else {
                score = len >= 1 ? score_out<R>(Cd + (double) csum + (double) Num<R>::log2(nrm)) : NINF;
                // a vanished or overflowed total is the per-utterance chain's business too
                if (len >= 1 && !(nrm > 0 && nrm < __builtin_inff())) wlo = 0;
            }
        }
    }
    const bool flagged = wlo < Rng<R>::lo || whi > Rng<R>::hi;
    if (flagged) atomicOr(&L.flag[n], 1);
    wg_barrier();
    if (w == 0) {
        const bool fl = L.flag[n] != 0;
        if (g == 0 && uv) flags[(BETA ? B : 0) + b] = fl ? 1 : 0;
        if (BETA) publish_scores_x16(O, (R *) O.full_scores, b, B, score, g == 0 && uv && !fl, lane);
        else if (O.full_scores_alpha && g == 0 && uv) ((R *) O.full_scores_alpha)[b] = score;
    }
}

}  // namespace
}  // namespace asg
