// torch_asg_amd/csrc/asg_small.hip -- gfx950 kernels for the small-alphabet ASG path
// (N <= 64 labels, S <= 64 target positions): ONE 64-lane wavefront per recursion chain.
//
// What each kernel replaces in the reference (paths under /root/reference/torch_asg/native/):
//   full_alpha_chain    fully_connected_lattice.cpp:9-29   (alpha recursion; no path_contrib tensor)
//   full_beta_chain     fully_connected_lattice.cpp:32-47,65-91 (beta recursion, length handling w/o roll_to_end)
//   aligned_alpha_chain force_aligned_lattice.cpp:84-111 + the gathers :15-82 / kernel.cu:7-226 fused in
//   aligned_beta_chain  force_aligned_lattice.cpp:113-154
//   bwd_small_kernel    fully_connected_lattice.cpp:49-63,93-105 + force_aligned_lattice.cpp:156-264,321-356
//                       (+ the atomicAdd scatter kernels force_aligned_lattice_kernel.cu:253-470)
//
// Design: the full-lattice step  alpha_t[i] = I_t[i] + LSE_j(Tr[i][j] + alpha_{t-1}[j])
// is evaluated as a row-normalised exp-domain mat-vec: lane i keeps row i of exp2(Tr2 - rowmax) in
// registers, the previous frame's p_j = exp2(alpha_hat_j) (max == 1) is broadcast through LDS (or
// v_readlane), N FMAs, one v_log_f32.  If a row sum falls below 1e-30 the lane recomputes that node with
// an exact max-shifted log-sum-exp, so the result is a true LSE for any input range.
// The aligned lattice stays in the log domain (2-term LSE per node) because its band structure makes
// per-frame dynamic range unbounded for tight alignments.
#include "asg_common.h"
#include "asg_kernels.h"

namespace asg {

namespace {

constexpr int kPF = 16;   // emission prefetch depth (frames), register ring

template <typename R> struct Vec4 { R x, y, z, w; };

__device__ __forceinline__ int clampi(int64_t v, int lo, int hi) {
    return v < lo ? lo : (v > hi ? hi : (int) v);
}

// s_i = sum_j e[j] * p_j with p_j living in lane j.
template <typename R, int NP, int MV>
__device__ __forceinline__ R matvec(const R (&e)[NP], R p, R *lds, int lane) {
    R s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (MV == 0) {
        // LDS broadcast: one ds_write_b32 per lane, then every lane reads the whole vector with
        // wide same-address (broadcast, conflict-free) reads.  A single wavefront owns `lds`, and the
        // LDS executes one wave's DS ops in order, so no barrier is needed -- only a compiler fence.
        lds[lane] = p;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
            Vec4<R> v = *reinterpret_cast<const Vec4<R> *>(lds + j);
            s0 = fma(e[j + 0], v.x, s0);
            s1 = fma(e[j + 1], v.y, s1);
            s2 = fma(e[j + 2], v.z, s2);
            s3 = fma(e[j + 3], v.w, s3);
        }
        __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
        for (int j = 0; j < NP; j += 4) {
            s0 = fma(e[j + 0], readlane(p, j + 0), s0);
            s1 = fma(e[j + 1], readlane(p, j + 1), s1);
            s2 = fma(e[j + 2], readlane(p, j + 2), s2);
            s3 = fma(e[j + 3], readlane(p, j + 3), s3);
        }
    }
    return (s0 + s1) + (s2 + s3);
}

// Exact log2-sum-exp2 over j of (trow[j*tstride]*log2e + v_j), v_j in lane j.  Rare path.
template <typename R>
__device__ __noinline__ R exact_lse_row(const R *trow, int64_t tstride, R v, int N, bool act) {
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    R mx = NINF;
    for (int j = 0; j < N; ++j) {
        R vj = readlane(v, j);
        R x = act ? trow[j * tstride] * L2E + vj : NINF;
        mx = fmax(mx, x);
    }
    R sm = 0;
    for (int j = 0; j < N; ++j) {
        R vj = readlane(v, j);
        R x = act ? trow[j * tstride] * L2E + vj : NINF;
        sm += (mx == NINF) ? R(0) : Num<R>::exp2(x - mx);
    }
    return (mx == NINF) ? NINF : mx + Num<R>::log2(sm);
}

// ------------------------------------------------------------------ full lattice, alpha
template <typename R, int NP, int MV, bool STORE>
__device__ void full_alpha_chain(const Problem &P, const State &W, const FwdOut &O, int b, R *lds) {
    constexpr int ROWS = (NP + 15) / 16;
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    const bool act = lane < N;
    const R *tr = (const R *) P.transition;
    const R *trow = tr + (act ? (int64_t) lane * P.ts0 : 0);

    R e[NP];
    R Ri = NINF;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        R v = (act && j < N) ? trow[(int64_t) j * P.ts1] * L2E : NINF;
        e[j] = v;
        Ri = fmax(Ri, v);
    }
    if (Ri == NINF) Ri = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) e[j] = Num<R>::exp2(e[j] - Ri);

    const R *in = (const R *) P.inputs + (int64_t) b * P.is1 + (act ? (int64_t) lane * P.is2 : 0);
    R *ah_out = (R *) W.ah + (int64_t) b * T * N + lane;
    R *msh_out = (R *) W.msh + (int64_t) b * T;

    R cur[kPF], nxt[kPF];
#pragma unroll
    for (int k = 0; k < kPF; ++k) cur[k] = (act && k < len) ? in[(int64_t) k * P.is0] * L2E : NINF;

    double C = 0.0;
    R ah = NINF;
    for (int tb = 0; tb < len; tb += kPF) {
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            int tt = tb + kPF + k;
            nxt[k] = (act && tt < len) ? in[(int64_t) tt * P.is0] * L2E : NINF;
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            const int t = tb + k;
            if (t >= len) break;
            R x;
            if (t == 0) {
                x = cur[k];
            } else {
                R p = Num<R>::exp2(ah);                      // lanes >= N hold -inf -> 0
                R s = matvec<R, NP, MV>(e, p, lds, lane);
                x = cur[k] + Ri + Num<R>::log2(s);
                bool bad = act && !(s >= Num<R>::tiny());
                if (__any(bad)) {
                    R ex = exact_lse_row<R>(trow, P.ts1, ah, N, act);
                    if (bad) x = cur[k] + ex;
                }
            }
            R m = wave_allmax<ROWS>(x);
            if (m == NINF) { ah = NINF; C = -__builtin_inf(); m = 0; }
            else { ah = x - m; C += (double) m; }
            if (STORE) {
                if (act) ah_out[(int64_t) t * N] = ah;
                if (lane == 0) msh_out[t] = m;
            }
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
    }
    if (O.full_scores_alpha) {
        R sm = wave_allsum<ROWS>(Num<R>::exp2(ah));
        double sc = (len >= 1) ? (C + (double) Num<R>::log2(sm)) * kLn2 : -__builtin_inf();
        if (lane == 0) ((R *) O.full_scores_alpha)[b] = (R) sc;
    }
}

// ------------------------------------------------------------------ full lattice, beta
template <typename R, int NP, int MV, bool STORE>
__device__ void full_beta_chain(const Problem &P, const State &W, const FwdOut &O, int b, R *lds) {
    constexpr int ROWS = (NP + 15) / 16;
    const int lane = threadIdx.x & 63;
    const int N = P.N, T = P.T;
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    const bool act = lane < N;
    const R *tr = (const R *) P.transition;
    const R *tcol = tr + (act ? (int64_t) lane * P.ts1 : 0);      // column `lane`: Tr[j][lane]

    R f[NP];
    R Ci = NINF;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        R v = (act && j < N) ? tcol[(int64_t) j * P.ts0] * L2E : NINF;
        f[j] = v;
        Ci = fmax(Ci, v);
    }
    if (Ci == NINF) Ci = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) f[j] = Num<R>::exp2(f[j] - Ci);

    const R *in = (const R *) P.inputs + (int64_t) b * P.is1 + (act ? (int64_t) lane * P.is2 : 0);
    R *bh_out = (R *) W.bh + (int64_t) b * T * N + lane;

    if (len < 1) {
        if (lane == 0) ((R *) O.full_scores)[b] = NINF;
        return;
    }
    // iteration n handles frame t = len-1-n: y = I2[t] + bh[t], then produces bh[t-1]
    R cur[kPF], nxt[kPF];
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        int t = len - 1 - k;
        cur[k] = (act && t >= 0) ? in[(int64_t) t * P.is0] * L2E : NINF;
    }
    double C = 0.0;
    R bh = act ? R(0) : NINF;
    if (STORE && act) bh_out[(int64_t) (len - 1) * N] = bh;
    for (int nb = 0; nb < len; nb += kPF) {
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            int t = len - 1 - (nb + kPF + k);
            nxt[k] = (act && t >= 0) ? in[(int64_t) t * P.is0] * L2E : NINF;
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            const int t = len - 1 - (nb + k);
            if (t < 0) break;
            R y = cur[k] + bh;
            R my = wave_allmax<ROWS>(y);
            if (t == 0) {
                R sm = wave_allsum<ROWS>((my == NINF) ? R(0) : Num<R>::exp2(y - my));
                double sc = (my == NINF) ? -__builtin_inf() : (C + (double) my + (double) Num<R>::log2(sm)) * kLn2;
                if (lane == 0) ((R *) O.full_scores)[b] = (R) sc;
                break;
            }
            if (my == NINF) {
                bh = NINF; C = -__builtin_inf();
            } else {
                R p = Num<R>::exp2(y - my);
                R s = matvec<R, NP, MV>(f, p, lds, lane);
                bh = Ci + Num<R>::log2(s);
                bool bad = act && !(s >= Num<R>::tiny());
                if (__any(bad)) {
                    R ex = exact_lse_row<R>(tcol, P.ts0, y, N, act);
                    if (bad) bh = ex - my;
                }
                if (!act) bh = NINF;
                C += (double) my;
            }
            if (STORE && act) bh_out[(int64_t) (t - 1) * N] = bh;
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
    }
}

// ------------------------------------------------------------------ aligned lattice
template <typename R>
struct AlignedSetup {
    int len, ol, tgt;
    bool act;
    R H2;      // Tr2[O_s][O_s]
    R Dprev;   // Tr2[O_s][O_{s-1}]   (edge s-1 -> s), 0 for s == 0
    R Dnext;   // Tr2[O_{s+1}][O_s]   (edge s -> s+1), 0 for s >= ol-1
    const R *in;   // &inputs[0][b][O_s]
};

template <typename R>
__device__ __forceinline__ AlignedSetup<R> aligned_setup(const Problem &P, int b, int lane) {
    AlignedSetup<R> A;
    const R L2E = Num<R>::log2e();
    A.len = P.in_len ? clampi(P.in_len[b], 0, P.T) : P.T;
    A.ol = P.tg_len ? clampi(P.tg_len[b], 0, P.S) : P.S;
    A.act = lane < A.ol;
    const int64_t *tg = P.targets + (int64_t) b * P.gs0;
    int cur = A.act ? clampi(tg[(int64_t) lane * P.gs1], 0, P.N - 1) : 0;
    int prv = (A.act && lane >= 1) ? clampi(tg[(int64_t) (lane - 1) * P.gs1], 0, P.N - 1) : 0;
    int nxt = (lane + 1 < A.ol) ? clampi(tg[(int64_t) (lane + 1) * P.gs1], 0, P.N - 1) : 0;
    const R *tr = (const R *) P.transition;
    A.tgt = cur;
    A.H2 = A.act ? tr[(int64_t) cur * P.ts0 + (int64_t) cur * P.ts1] * L2E : R(0);
    A.Dprev = (A.act && lane >= 1) ? tr[(int64_t) cur * P.ts0 + (int64_t) prv * P.ts1] * L2E : R(0);
    A.Dnext = (lane + 1 < A.ol) ? tr[(int64_t) nxt * P.ts0 + (int64_t) cur * P.ts1] * L2E : R(0);
    A.in = (const R *) P.inputs + (int64_t) b * P.is1 + (int64_t) cur * P.is2;
    return A;
}

template <typename R, bool STORE>
__device__ void aligned_alpha_chain(const Problem &P, const State &W, const FwdOut &O, int b) {
    const int lane = threadIdx.x & 63;
    const int T = P.T, S = P.S;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    const AlignedSetup<R> A = aligned_setup<R>(P, b, lane);
    const int len = A.len;
    R *ab_out = (R *) W.ab + (int64_t) b * T * S + lane;
    const bool st = STORE && lane < S;

    R cur[kPF], nxt[kPF];
#pragma unroll
    for (int k = 0; k < kPF; ++k) cur[k] = (A.act && k < len) ? A.in[(int64_t) k * P.is0] * L2E : NINF;
    double C = 0.0;
    R ab = NINF;
    for (int tb = 0; tb < len; tb += kPF) {
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            int tt = tb + kPF + k;
            nxt[k] = (A.act && tt < len) ? A.in[(int64_t) tt * P.is0] * L2E : NINF;
        }
        if (tb > 0) {   // renormalise once per block: log domain is offset-free, this only bounds magnitudes
            R m = wave_allmax<4>(ab);
            if (m != NINF) { ab -= m; C += (double) m; }
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            const int t = tb + k;
            if (t >= len) break;
            if (t == 0) {
                ab = (lane == 0) ? cur[k] : NINF;
            } else {
                R left = from_prev_lane<R>(ab, NINF);
                ab = cur[k] + lse2<R>(ab + A.H2, left + A.Dprev);
            }
            if (st) ab_out[(int64_t) t * S] = ab;
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
    }
    if (O.aligned_scores_alpha) {
        R last = (A.ol >= 1 && len >= 1) ? readlane(ab, A.ol - 1) : NINF;
        if (lane == 0) ((R *) O.aligned_scores_alpha)[b] = (R) ((C + (double) last) * kLn2);
    }
}

template <typename R, bool STORE>
__device__ void aligned_beta_chain(const Problem &P, const State &W, const FwdOut &O, int b) {
    const int lane = threadIdx.x & 63;
    const int T = P.T, S = P.S;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    const AlignedSetup<R> A = aligned_setup<R>(P, b, lane);
    const int len = A.len;
    R *bb_out = (R *) W.bb + (int64_t) b * T * S + lane;
    const bool st = STORE && lane < S;
    if (len < 1 || A.ol < 1) {
        if (lane == 0) ((R *) O.aligned_scores)[b] = NINF;
        return;
    }
    R cur[kPF], nxt[kPF];
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
        int t = len - 1 - k;
        cur[k] = (A.act && t >= 0) ? A.in[(int64_t) t * P.is0] * L2E : NINF;
    }
    double C = 0.0;
    R bb = (lane == A.ol - 1) ? R(0) : NINF;
    if (st) bb_out[(int64_t) (len - 1) * S] = bb;
    for (int nb = 0; nb < len; nb += kPF) {
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            int t = len - 1 - (nb + kPF + k);
            nxt[k] = (A.act && t >= 0) ? A.in[(int64_t) t * P.is0] * L2E : NINF;
        }
        if (nb > 0) {
            R m = wave_allmax<4>(bb);
            if (m != NINF) { bb -= m; C += (double) m; }
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
            const int t = len - 1 - (nb + k);
            if (t < 0) break;
            R y = cur[k] + bb;
            if (t == 0) {
                R y0 = readlane(y, 0);
                if (lane == 0) ((R *) O.aligned_scores)[b] = (R) ((C + (double) y0) * kLn2);
                break;
            }
            R right = from_next_lane<R>(y, NINF);
            bb = lse2<R>(A.H2 + y, A.Dnext + right);
            if (st) bb_out[(int64_t) (t - 1) * S] = bb;
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) cur[k] = nxt[k];
    }
}

// ------------------------------------------------------------------ forward kernel
// grid = (B, popcount(chain_mask)), block = 64.  blockIdx.y walks the set bits of chain_mask
// low to high, so the long full-lattice chains are dispatched first.
template <typename R, int NP, int MV, bool STORE>
__global__ void __launch_bounds__(64) fwd_small_kernel(Problem P, State W, FwdOut O, int chain_mask) {
    __shared__ __attribute__((aligned(16))) R lds[64];
    int which = 0, seen = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (chain_mask & (1 << c)) {
            if (seen == (int) blockIdx.y) which = 1 << c;
            ++seen;
        }
    }
    const int b = blockIdx.x;
    if (which == kFullAlpha) full_alpha_chain<R, NP, MV, STORE>(P, W, O, b, lds);
    else if (which == kFullBeta) full_beta_chain<R, NP, MV, STORE>(P, W, O, b, lds);
    else if (which == kAlignedAlpha) aligned_alpha_chain<R, STORE>(P, W, O, b);
    else if (which == kAlignedBeta) aligned_beta_chain<R, STORE>(P, W, O, b);
}

// ------------------------------------------------------------------ backward (gradient assembly)
// grid = (B, nchunks), block = 256 (4 waves).  Wave w of chunk c owns frames t = c*chunk + w, +4, ...
// Per frame: full-lattice posterior -> grad_inputs, outer-product accumulation of the transition
// gradient in registers (lane i holds row i), aligned posterior scattered back to labels with
// fixed-point LDS adds (deterministic), horizontal/diagonal edge posteriors accumulated per lane.
// Output: grad_inputs rows for its frames, one partial [N][N] tile per workgroup.
template <typename R, int NP>
__global__ void __launch_bounds__(256) bwd_small_kernel(Problem P, State W, BwdArgs A, int parts) {
    constexpr int ROWS = (NP + 15) / 16;
    __shared__ __attribute__((aligned(16))) R pbuf[4][64];
    __shared__ unsigned long long fxI[4][64];
    __shared__ unsigned long long fxT[64 * 64];
    __shared__ R tileF[64 * NP];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int N = P.N, T = P.T, S = P.S;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e();
    const bool do_full = parts & 1, do_ali = parts & 2;
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const bool act = lane < N;

    for (int k = threadIdx.x; k < N * N; k += 256) fxT[k] = 0;
    fxI[wave][lane] = 0;

    const R gf = do_full ? ((const R *) A.grad_full)[b] : R(0);
    const R ga = do_ali ? ((const R *) A.grad_aligned)[b] : R(0);

    // full: row i of exp2(Tr2 - rowmax)
    const R *tr = (const R *) P.transition;
    const R *trow = tr + (act ? (int64_t) lane * P.ts0 : 0);
    R e[NP];
    R Ri = NINF;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        R v = (act && j < N) ? trow[(int64_t) j * P.ts1] * L2E : NINF;
        e[j] = v;
        Ri = fmax(Ri, v);
    }
    if (Ri == NINF) Ri = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) e[j] = Num<R>::exp2(e[j] - Ri);
    R acc[NP], accx[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { acc[j] = 0; accx[j] = 0; }

    const AlignedSetup<R> AS = aligned_setup<R>(P, b, lane);
    R accH = 0, accD = 0;    // unscaled edge posteriors: stay on s ; arrive at s from s-1

    const R *in = (const R *) P.inputs + (int64_t) b * P.is1 + (act ? (int64_t) lane * P.is2 : 0);
    const R *ahp = (const R *) W.ah + (int64_t) b * T * N + lane;
    const R *bhp = (const R *) W.bh + (int64_t) b * T * N + lane;
    const R *mshp = (const R *) W.msh + (int64_t) b * T;
    const R *abp = (const R *) W.ab + (int64_t) b * T * S + lane;
    const R *bbp = (const R *) W.bb + (int64_t) b * T * S + lane;
    R *gin = (R *) A.grad_inputs + (int64_t) b * N + lane;
    __syncthreads();

    const int t0 = chunk * A.chunk;
    const int t1 = min(T, t0 + A.chunk);
    for (int t = t0 + wave; t < t1; t += 4) {
        R gi = 0;
        if (t < len) {
            if (do_full) {
                R ahv = act ? ahp[(int64_t) t * N] : NINF;
                R bhv = act ? bhp[(int64_t) t * N] : NINF;
                R gam = ahv + bhv;
                R mg = wave_allmax<ROWS>(gam);
                R w = (mg == NINF) ? R(0) : Num<R>::exp2(gam - mg);
                R Z = wave_allsum<ROWS>(w);
                gi = (Z > 0) ? gf * (w / Z) : R(0);
                if (t >= 1) {
                    R ahprev = act ? ahp[(int64_t) (t - 1) * N] : NINF;
                    R p = Num<R>::exp2(ahprev);
                    R i2 = act ? in[(int64_t) t * P.is0] * L2E : R(0);
                    R ls = ahv + mshp[t] - i2 - Ri;           // log2 of this row's mat-vec sum in the forward pass
                    bool live = act && gi != R(0);
                    bool bad = live && !(ls >= Num<R>::ls_floor());
                    R u = (live && !bad) ? gi * Num<R>::exp2(-ls) : R(0);
                    if (__any(bad)) {
                        // exact rare path (the forward pass re-did this node with an exact LSE):
                        // xi[i][j] = gi * exp2(Tr2[i][j] + ah_{t-1}[j] - (ah_t[i] + m_t - I2_t[i])), kept in accx (not scaled by e[j])
                        R base = ahv + mshp[t] - i2;
                        for (int j = 0; j < N; ++j) {
                            R aj = readlane(ahprev, j);
                            R x = bad ? gi * Num<R>::exp2(trow[(int64_t) j * P.ts1] * L2E + aj - base) : R(0);
#pragma unroll
                            for (int q = 0; q < NP; ++q) accx[q] += (q == j) ? x : R(0);
                        }
                    }
                    // acc[j] += u * p_j
                    R *lds = pbuf[wave];
                    lds[lane] = p;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int j = 0; j < NP; j += 4) {
                        Vec4<R> v = *reinterpret_cast<const Vec4<R> *>(lds + j);
                        acc[j + 0] = fma(u, v.x, acc[j + 0]);
                        acc[j + 1] = fma(u, v.y, acc[j + 1]);
                        acc[j + 2] = fma(u, v.z, acc[j + 2]);
                        acc[j + 3] = fma(u, v.w, acc[j + 3]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (do_ali) {
                const bool sl = lane < S;
                R abv = sl ? abp[(int64_t) t * S] : NINF;
                R bbv = sl ? bbp[(int64_t) t * S] : NINF;
                R gam = abv + bbv;
                R mg = wave_allmax<4>(gam);
                R w = (mg == NINF) ? R(0) : Num<R>::exp2(gam - mg);
                R Z = wave_allsum<4>(w);
                R post = (Z > 0) ? w / Z : R(0);            // unscaled state posterior, 0 for s >= ol
                if (AS.act && post != R(0))
                    atomicAdd(&fxI[wave][AS.tgt], to_fix<R>(post));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                unsigned long long fv = fxI[wave][lane];
                if (fv != 0) {
                    gi += ga * from_fix<R>(fv);
                    fxI[wave][lane] = 0;
                }
                __builtin_amdgcn_wave_barrier();
                if (t >= 1) {
                    R abprev = sl ? abp[(int64_t) (t - 1) * S] : NINF;
                    R left = from_prev_lane<R>(abprev, NINF);
                    R pc0 = abprev + AS.H2, pc1 = left + AS.Dprev;
                    R l = lse2<R>(pc0, pc1);
                    R hori = (pc0 == NINF) ? R(0) : Num<R>::exp2(pc0 - l);
                    R diag = (pc1 == NINF) ? R(0) : Num<R>::exp2(pc1 - l);
                    accH += post * hori;
                    accD += post * diag;
                }
            }
        }
        if (act) gin[(int64_t) t * P.B * N] = gi;
    }

    // ---- epilogue: one partial [N][N] tile per workgroup
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[j] = fma(acc[j], e[j], accx[j]);
    for (int w = 0; w < 4; ++w) {
        if (wave == w && act) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                R prev = (w == 0) ? R(0) : tileF[lane * NP + j];
                tileF[lane * NP + j] = prev + acc[j];
            }
        }
        __syncthreads();
    }
    if (do_ali && AS.act) {
        if (accH != R(0)) atomicAdd(&fxT[AS.tgt * N + AS.tgt], to_fix<R>(accH));
        if (lane >= 1 && accD != R(0)) {
            const int64_t *tg = P.targets + (int64_t) b * P.gs0;
            int prv = clampi(tg[(int64_t) (lane - 1) * P.gs1], 0, N - 1);
            atomicAdd(&fxT[AS.tgt * N + prv], to_fix<R>(accD));
        }
    }
    __syncthreads();
    R *tile_out = (R *) A.scratch + ((int64_t) b * A.nchunks + chunk) * N * N;
    for (int k = threadIdx.x; k < N * N; k += 256) {
        int i = k / N, j = k - i * N;
        R v = do_full ? tileF[i * NP + j] : R(0);
        unsigned long long fv = fxT[k];
        if (fv != 0) v += ga * from_fix<R>(fv);
        tile_out[k] = v;
    }
}

// sum G partial tiles in a fixed order -> deterministic grad_transition
template <typename R>
__global__ void __launch_bounds__(256) reduce_tiles_kernel(const R *tiles, int G, int n, R *out) {
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    R s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int g = 0;
    for (; g + 3 < G; g += 4) {
        s0 += tiles[(int64_t) (g + 0) * n + k];
        s1 += tiles[(int64_t) (g + 1) * n + k];
        s2 += tiles[(int64_t) (g + 2) * n + k];
        s3 += tiles[(int64_t) (g + 3) * n + k];
    }
    for (; g < G; ++g) s0 += tiles[(int64_t) g * n + k];
    out[k] = (s0 + s1) + (s2 + s3);
}

template <typename R, int NP, int MV>
hipError_t launch_fwd_np(const Problem &P, const State &W, const FwdOut &O, int mask, bool store, hipStream_t st) {
    dim3 grid(P.B, __builtin_popcount(mask)), block(64);
    if (store) hipLaunchKernelGGL((fwd_small_kernel<R, NP, MV, true>), grid, block, 0, st, P, W, O, mask);
    else hipLaunchKernelGGL((fwd_small_kernel<R, NP, MV, false>), grid, block, 0, st, P, W, O, mask);
    return hipGetLastError();
}

template <typename R, int MV>
hipError_t launch_fwd_mv(const Problem &P, const State &W, const FwdOut &O, int mask, bool store, hipStream_t st) {
    const int N = P.N;
    if (!(mask & (kFullAlpha | kFullBeta)) || N <= 8) return launch_fwd_np<R, 8, MV>(P, W, O, mask, store, st);
    if (N <= 16) return launch_fwd_np<R, 16, MV>(P, W, O, mask, store, st);
    if (N <= 24) return launch_fwd_np<R, 24, MV>(P, W, O, mask, store, st);
    if (N <= 32) return launch_fwd_np<R, 32, MV>(P, W, O, mask, store, st);
    if (N <= 40) return launch_fwd_np<R, 40, MV>(P, W, O, mask, store, st);
    if (N <= 48) return launch_fwd_np<R, 48, MV>(P, W, O, mask, store, st);
    if (N <= 56) return launch_fwd_np<R, 56, MV>(P, W, O, mask, store, st);
    return launch_fwd_np<R, 64, MV>(P, W, O, mask, store, st);
}

template <typename R, int NP>
hipError_t launch_bwd_np(const Problem &P, const State &W, const BwdArgs &A, int parts, hipStream_t st) {
    dim3 grid(P.B, A.nchunks), block(256);
    hipLaunchKernelGGL((bwd_small_kernel<R, NP>), grid, block, 0, st, P, W, A, parts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = P.N * P.N, G = P.B * A.nchunks;
    hipLaunchKernelGGL((reduce_tiles_kernel<R>), dim3((n + 255) / 256), dim3(256), 0, st,
                       (const R *) A.scratch, G, n, (R *) A.grad_transition);
    return hipGetLastError();
}

}  // namespace

template <typename R>
hipError_t launch_fwd_small(const Problem &P, const State &W, const FwdOut &O, int chain_mask, bool store,
                            int matvec_variant, hipStream_t stream) {
    if (chain_mask == 0) return hipSuccess;
    if (matvec_variant == 1) return launch_fwd_mv<R, 1>(P, W, O, chain_mask, store, stream);
    return launch_fwd_mv<R, 0>(P, W, O, chain_mask, store, stream);
}

template <typename R>
hipError_t launch_bwd_small(const Problem &P, const State &W, const BwdArgs &A, int parts, hipStream_t stream) {
    const int N = P.N;
    if (N <= 8) return launch_bwd_np<R, 8>(P, W, A, parts, stream);
    if (N <= 16) return launch_bwd_np<R, 16>(P, W, A, parts, stream);
    if (N <= 24) return launch_bwd_np<R, 24>(P, W, A, parts, stream);
    if (N <= 32) return launch_bwd_np<R, 32>(P, W, A, parts, stream);
    if (N <= 40) return launch_bwd_np<R, 40>(P, W, A, parts, stream);
    if (N <= 48) return launch_bwd_np<R, 48>(P, W, A, parts, stream);
    if (N <= 56) return launch_bwd_np<R, 56>(P, W, A, parts, stream);
    return launch_bwd_np<R, 64>(P, W, A, parts, stream);
}

size_t bwd_scratch_bytes_small(int elem, int T, int B, int N, int S, int *chunk, int *nchunks) {
    // aim for ~512 workgroups (2 per CU) but at least 16 frames per workgroup
    int nch = (512 + B - 1) / B;
    if (nch < 1) nch = 1;
    int ch = (T + nch - 1) / nch;
    if (ch < 16) ch = 16;
    ch = (ch + 3) / 4 * 4;
    nch = (T + ch - 1) / ch;
    if (nch < 1) nch = 1;
    if (chunk) *chunk = ch;
    if (nchunks) *nchunks = nch;
    return (size_t) B * nch * N * N * elem;
}

template hipError_t launch_fwd_small<float>(const Problem &, const State &, const FwdOut &, int, bool, int, hipStream_t);
template hipError_t launch_fwd_small<double>(const Problem &, const State &, const FwdOut &, int, bool, int, hipStream_t);
template hipError_t launch_bwd_small<float>(const Problem &, const State &, const BwdArgs &, int, hipStream_t);
template hipError_t launch_bwd_small<double>(const Problem &, const State &, const BwdArgs &, int, hipStream_t);

}  // namespace asg
