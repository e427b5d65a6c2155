// torch_asg_amd/csrc/asg_assemble.h -- gfx950 device code of the non-recursive gradient assembly (small-alphabet path)
// and the fixed-order tile / loss reductions.  Replaces fully_connected_lattice.cpp:49-63,93-105 and
// force_aligned_lattice.cpp:156-264,321-356 (+ the atomicAdd scatter kernels force_aligned_lattice_kernel.cu:253-470) of
// /root/reference/torch_asg/native/.  Included by asg_bwd_f32/f64.hip (stand-alone assembly kernels) and by
// asg_fused.hip (exact in-launch fallback of the fused forward+assembly kernel).
#pragma once
#include "asg_chains.h"

namespace asg {
namespace {

// ------------------------------------------------------------------ backward (gradient assembly)
// grid = (B, nchunks), block = 256 (4 waves).  Wave w of chunk c owns frames t = c*chunk + w, +NW, ...
// Per frame (non-recursive, every frame independent):
//   full:    posterior_i = softmax_i(alpha_hat + beta_hat)                    -> grad_inputs row
//            p_j = exp2(alpha_hat_{t-1}[j] - max), s_i = sum_j E[i][j] p_j    (row sums recomputed here, so the
//            forward pass has nothing to save but alpha_hat / beta_hat), u_i = g * posterior_i / s_i,
//            acc[i][j] += u_i * p_j   (lane i keeps row i in registers; scaled by E[i][j] once at the end)
//   aligned: posterior_s = softmax_s(alpha_bar + beta_bar), scattered back to labels with fixed-point
//            LDS adds (integer adds commute -> deterministic), stay/advance edge posteriors per lane.
// Rows whose recomputed sum is outside the safe range are skipped on the fast path (sticky flag) and handled
// by an exact second pass over the wave's frames into a fixed-point LDS tile -- rare, off the fast path.
// Output: grad_inputs rows for its frames, one partial [N][N] tile per workgroup.

// LDS of one assembly workgroup of NW wavefronts.
template <typename R, int NP, int NW>
struct AssembleLds {
    __attribute__((aligned(16))) R pbuf[NW][64];
    typename FrameFix<R>::T fxI[NW][64];
    unsigned long long fxT[NP * NP];      // aligned edge posteriors (unscaled)
    unsigned long long fxX[NP * NP];      // exact-path full-lattice edge posteriors (unscaled)
    __attribute__((aligned(16))) R tileF[64 * NP];
};

// Gradient assembly of frames [chunk*A.chunk, (chunk+1)*A.chunk) of utterance b by the NW wavefronts of the calling
// workgroup (all of its threads must call: the body contains workgroup barriers); writes the grad_inputs rows of
// those frames and ONE partial [N][N] tile to tile_out.
template <typename R, int NP, int NW>
__device__ __forceinline__ void assemble_frames(const Problem &P, const State &W, const BwdArgs &A, int parts, int b,
                                                int chunk, R *tile_out, AssembleLds<R, NP, NW> &LS) {
    auto &pbuf = LS.pbuf;
    auto &fxI = LS.fxI;
    auto &fxT = LS.fxT;
    auto &fxX = LS.fxX;
    auto &tileF = LS.tileF;
    constexpr int NT = NW * 64;

    const int lane = threadIdx.x & 63;
    // wave index made provably uniform: otherwise every frame index, pointer and store offset derived from it is
    // treated as divergent (EXEC-masked loop control, waterfall loop around the buffer store)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = P.N, T = P.T, S = P.S;
    const R NINF = Num<R>::ninf(), L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const bool do_full = parts & 1, do_ali = parts & 2;
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const int ol = (do_ali && P.targets) ? (P.tg_len ? clampi(P.tg_len[b], 0, S) : S) : 0;
    const bool act = lane < N, sl = lane < S, sact = lane < ol;
    const int lc = act ? lane : 0, ls_ = sl ? lane : 0;

    const R *ahp = (const R *) W.ah + (int64_t) b * T * N + lc;
    const R *bhp = (const R *) W.bh + (int64_t) b * T * N + lc;
    // state rows through buffer loads: lane offset in a VGPR, frame offset in an SGPR (no per-lane 64-bit address math)
    __amdgpu_buffer_rsrc_t r_ah = make_rsrc((R *) W.ah + (int64_t) b * T * N, (unsigned) T * (unsigned) N * (unsigned) sizeof(R));
    __amdgpu_buffer_rsrc_t r_bh = make_rsrc((R *) W.bh + (int64_t) b * T * N, (unsigned) T * (unsigned) N * (unsigned) sizeof(R));
    __amdgpu_buffer_rsrc_t r_ab = make_rsrc((R *) W.ab + (int64_t) b * T * S, (unsigned) T * (unsigned) S * (unsigned) sizeof(R));
    __amdgpu_buffer_rsrc_t r_bb = make_rsrc((R *) W.bb + (int64_t) b * T * S, (unsigned) T * (unsigned) S * (unsigned) sizeof(R));
    const unsigned vN = (unsigned) lc * (unsigned) sizeof(R), vS = (unsigned) ls_ * (unsigned) sizeof(R);
    const unsigned rbN = (unsigned) N * (unsigned) sizeof(R), rbS = (unsigned) S * (unsigned) sizeof(R);
    const int t0 = chunk * A.chunk;
    const int t1 = min(T, t0 + A.chunk);
    // software prefetch: the six state values of the NEXT frame are loaded before the current one is processed; the
    // first frame's are issued before anything else so that their latency hides under the rest of the prologue
    R n_ah, n_bh, n_ahp, n_ab, n_bb, n_abp;
    {
        const int tq = min(t0 + wave, T - 1), tqp = tq >= 1 ? tq - 1 : 0;
        n_ah = buf_load<R>(r_ah, vN, (unsigned) tq * rbN); n_bh = buf_load<R>(r_bh, vN, (unsigned) tq * rbN);
        n_ahp = buf_load<R>(r_ah, vN, (unsigned) tqp * rbN);
        n_ab = buf_load<R>(r_ab, vS, (unsigned) tq * rbS); n_bb = buf_load<R>(r_bb, vS, (unsigned) tq * rbS);
        n_abp = buf_load<R>(r_ab, vS, (unsigned) tqp * rbS);
    }
    // ---- prologue: everything below is ONE round of independent loads
    const R g0 = A.unit_grad ? (R) A.gscale
                             : (A.grad_full ? (R) ((double) ((const R *) A.grad_full)[(int64_t) b * A.gstride] * A.gscale) : R(0));
    const R gf = do_full ? g0 : R(0);
    const R ga = do_ali ? (A.neg_aligned ? -g0
                                         : (R) ((double) ((const R *) A.grad_aligned)[(int64_t) b * A.gstride] * A.gscale))
                        : R(0);
    V2<R> e2[NP / 2];
    if (do_full) {
        const V4<R> *erow = reinterpret_cast<const V4<R> *>((const R *) W.ehat + (int64_t) lc * W.npad);
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) {
            V4<R> v = erow[j];
            e2[2 * j] = v.xy;
            e2[2 * j + 1] = v.zw;
        }
        if (!act) {
#pragma unroll
            for (int j = 0; j < NP / 2; ++j) e2[j] = V2<R>{0, 0};
        }
    }
    V2<R> hd = {0, 0};
    int2 tp = {0, 0};
    if (do_ali) {
        hd = reinterpret_cast<const V2<R> *>(W.asu)[(int64_t) b * S + ls_];
        tp = reinterpret_cast<const int2 *>(W.asi)[(int64_t) b * S + ls_];
    }
    const R H2 = hd.x, Dprev = hd.y;
    const int tgt = tp.x, prv = tp.y;

    for (int k = threadIdx.x; k < N * N; k += NT) { fxT[k] = 0; fxX[k] = 0; }
    fxI[wave][lane] = 0;

    V2<R> acc[NP / 2];
#pragma unroll
    for (int j = 0; j < NP / 2; ++j) acc[j] = V2<R>{0, 0};
    R accH = 0, accD = 0;    // unscaled edge posteriors: stay on s ; arrive at s from s-1
    bool any_bad = false;

    __amdgpu_buffer_rsrc_t rs_g = make_rsrc((R *) A.grad_inputs + (int64_t) b * N,
                                            (unsigned) ((int64_t) (T - 1) * P.B * N + N) * (unsigned) sizeof(R));
    const unsigned voff = act ? (unsigned) lane * sizeof(R) : kOobOffset;
    const unsigned grow_bytes = (unsigned) P.B * N * sizeof(R);
    __syncthreads();

    for (int t = t0 + wave; t < t1; t += NW) {
        R gi = 0;
        const R c_ah = n_ah, c_bh = n_bh, c_ahp = n_ahp, c_ab = n_ab, c_bb = n_bb, c_abp = n_abp;
        {
            const int tq = min(t + NW, T - 1), tqp = tq >= 1 ? tq - 1 : 0;
            n_ah = buf_load<R>(r_ah, vN, (unsigned) tq * rbN); n_bh = buf_load<R>(r_bh, vN, (unsigned) tq * rbN);
            n_ahp = buf_load<R>(r_ah, vN, (unsigned) tqp * rbN);
            n_ab = buf_load<R>(r_ab, vS, (unsigned) tq * rbS); n_bb = buf_load<R>(r_bb, vS, (unsigned) tq * rbS);
            n_abp = buf_load<R>(r_ab, vS, (unsigned) tqp * rbS);
        }
        if (t < len) {
            // the three maxima (full gamma, previous alpha, aligned gamma) in one interleaved reduction pass
            R gam = act ? c_ah + c_bh : NINF;
            R ahprev = act ? c_ahp : NINF;
            R gam2 = sl ? c_ab + c_bb : LZ;
            R abprev = sl ? c_abp : LZ;
            R mg = gam, mg2 = gam2;
            wave_allmax2(mg, mg2);
            mg = fmax(mg, LZ);
            R w = do_full ? Num<R>::exp2(gam - mg) : R(0);
            R w2 = (do_ali && mg2 > R(-1e29)) ? Num<R>::exp2(gam2 - mg2) : R(0);   // infeasible alignment -> no posterior
            // the forward pass stores alpha_hat relative to an offset that keeps the frame's L1 norm near 1, so it is
            // exponentiated as is (no third reduction); a frame that underflows anyway fails the `ok` test below and
            // goes through the exact pass
            R p = Num<R>::exp2(ahprev);
            R *lds = pbuf[wave];
            if (do_full && t >= 1) {
                lds[lane] = p;
                __builtin_amdgcn_wave_barrier();
            }
            R Z = w, Z2 = w2;
            wave_allsum2(Z, Z2);
            // v_rcp (1 ulp) instead of the ~10-instruction IEEE division: far inside the 1e-4 budget
            R post2 = (Z2 > 0) ? w2 * Num<R>::rcp(Z2) : R(0);   // unscaled aligned state posterior, 0 for s >= ol
            gi = (Z > 0) ? gf * (w * Num<R>::rcp(Z)) : R(0);
            if (do_full && t >= 1) {
                V4<R> pv[NP / 4];
#pragma unroll
                for (int j = 0; j < NP / 4; ++j) pv[j] = *reinterpret_cast<const V4<R> *>(lds + 4 * j);
                __builtin_amdgcn_sched_barrier(0);
                V2<R> a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
                for (int j = 0; j < NP / 4; ++j) {
                    a0 = fma2<R>(e2[2 * j], pv[j].xy, a0);
                    a1 = fma2<R>(e2[2 * j + 1], pv[j].zw, a1);
                }
                V2<R> a = a0 + a1;
                R sden = a.x + a.y;                    // row sum of the forward mat-vec (up to the common scale of p)
                bool ok = fabs(Num<R>::log2(sden)) < Num<R>::lg_limit();
                any_bad |= (gi != R(0)) && !ok;
                R u = ok ? gi * Num<R>::rcp(sden) : R(0);
                const V2<R> u2 = {u, u};
#pragma unroll
                for (int j = 0; j < NP / 4; ++j) {
                    acc[2 * j] = fma2<R>(u2, pv[j].xy, acc[2 * j]);
                    acc[2 * j + 1] = fma2<R>(u2, pv[j].zw, acc[2 * j + 1]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (do_ali) {
                // unconditional (post2 is 0 on lanes >= ol, and adding 0 is harmless): no EXEC juggling per frame
                atomicAdd(&fxI[wave][tgt], FrameFix<R>::to(post2));
                __builtin_amdgcn_wave_barrier();
                if (t >= 1) {
                    R pc0 = abprev + H2;
                    R pc1 = prev_lane_or_zero<R>(abprev) + Dprev;
                    R l = lse2<R>(pc0, pc1);
                    accH += post2 * Num<R>::exp2(pc0 - l);
                    accD += post2 * Num<R>::exp2(pc1 - l);
                }
                const typename FrameFix<R>::T fv = fxI[wave][lane];
                gi += ga * FrameFix<R>::from(fv);
                fxI[wave][lane] = 0;
                __builtin_amdgcn_wave_barrier();
            }
        }
        buf_store(gi, rs_g, voff, (unsigned) t * grow_bytes);
    }

    // ---- rare exact pass: rows whose recomputed sum was unusable (forward took its exact path there too)
    if (do_full && __any(any_bad)) {
        const R *trow = (const R *) P.transition + (int64_t) lc * P.ts0;
        // same frame ownership as the fast loop (t = t0 + wave + NW*k); frame 0 has no incoming transition
        for (int t = (t0 + wave == 0) ? NW : t0 + wave; t < min(t1, len); t += NW) {
            R ahv = act ? ahp[(int64_t) t * N] : NINF, bhv = act ? bhp[(int64_t) t * N] : NINF;
            R ahprev = act ? ahp[(int64_t) (t - 1) * N] : NINF;
            R gam = ahv + bhv;
            R mg = fmax(wave_allmax(gam), LZ);
            R w = Num<R>::exp2(gam - mg);
            R Z = wave_allsum(w);
            R post = (Z > 0) ? w * Num<R>::rcp(Z) : R(0);
            // the SAME row sums, bit for bit, as the fast path computed (same operands, same order), so that "bad"
            // here is exactly the set of rows the fast path skipped
            R p = Num<R>::exp2(ahprev);
            R *lds = pbuf[wave];
            lds[lane] = p;
            __builtin_amdgcn_wave_barrier();
            V2<R> a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) {
                const V4<R> pvj = *reinterpret_cast<const V4<R> *>(lds + 4 * j);
                a0 = fma2<R>(e2[2 * j], pvj.xy, a0);
                a1 = fma2<R>(e2[2 * j + 1], pvj.zw, a1);
            }
            __builtin_amdgcn_wave_barrier();
            const V2<R> a = a0 + a1;
            const R sden = a.x + a.y;
            const R gi_f = (Z > 0) ? gf * (w * Num<R>::rcp(Z)) : R(0);
            const bool ok = fabs(Num<R>::log2(sden)) < Num<R>::lg_limit();
            bool bad = act && gi_f != R(0) && !ok;
            if (__any(bad)) {
                R lse = exact_lse_row<R>(trow, P.ts1, ahprev, N, act);
                for (int j = 0; j < N; ++j) {
                    R aj = readlane(ahprev, j);
                    R x = bad ? post * Num<R>::exp2(trow[(int64_t) j * P.ts1] * L2E + aj - lse) : R(0);
                    if (x == x && x != R(0)) atomicAdd(&fxX[lane * N + j], to_fix<R>(x));
                }
            }
        }
    }

    // ---- epilogue: one partial [N][N] tile per workgroup
#pragma unroll
    for (int j = 0; j < NP / 2; ++j) acc[j] = acc[j] * e2[j];
    for (int w = 0; w < NW; ++w) {
        if (wave == w && act) {
#pragma unroll
            for (int j = 0; j < NP / 2; ++j) {
                V2<R> *dst = reinterpret_cast<V2<R> *>(&tileF[lane * NP + 2 * j]);
                V2<R> prev = (w == 0) ? V2<R>{0, 0} : *dst;
                *dst = prev + acc[j];
            }
        }
        __syncthreads();
    }
    if (do_ali && sact) {
        if (accH != R(0)) atomicAdd(&fxT[tgt * N + tgt], to_fix<R>(accH));
        if (lane >= 1 && accD != R(0)) atomicAdd(&fxT[tgt * N + prv], to_fix<R>(accD));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N * N; k += NT) {
        int i = k / N, j = k - i * N;
        R v = do_full ? tileF[i * NP + j] : R(0);
        unsigned long long fv = fxT[k], fx = fxX[k];
        if (fv != 0) v += ga * from_fix<R>(fv);
        if (fx != 0) v += gf * from_fix<R>(fx);
        tile_out[k] = v;
    }
}

template <typename R, int NP>
__global__ void __launch_bounds__(256) bwd_small_kernel(Problem P, State W, BwdArgs A, int parts) {
    __shared__ AssembleLds<R, NP, 4> S;
    const int b = blockIdx.x, chunk = blockIdx.y;
    R *tile_out = (R *) A.scratch + ((int64_t) b * A.nchunks + chunk) * P.N * P.N;
    assemble_frames<R, NP, 4>(P, W, A, parts, b, chunk, tile_out, S);
}

// Sum G partial tiles in a fixed order -> deterministic grad_transition.
// block = 1024 threads = 32 elements x 32 tile-groups; thread (e, grp) sums tiles grp, grp+32, ... with 16
// independent accumulators (16 loads in flight: the kernel is pure L2 latency, so the 512 tiles of cfg 3 take ONE
// round of loads per thread), then a fixed-order LDS combine over the 32 groups.
constexpr int kRedGroups = 32;
template <typename R>
__global__ void __launch_bounds__(1024) reduce_tiles_kernel(const R *tiles, int G, int n, R *out) {
    __shared__ R part[kRedGroups][32];
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int k = min(blockIdx.x * 32 + e, n - 1);
    R acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0;
    int g = grp;
    for (; g + kRedGroups * 15 < G; g += kRedGroups * 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += tiles[(int64_t) (g + kRedGroups * q) * n + k];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        int gg = g + kRedGroups * q;
        R v = tiles[(int64_t) min(gg, G - 1) * n + k];        // unconditional load, masked add
        acc[q] += (gg < G) ? v : R(0);
    }
    R s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[q];
    part[grp][e] = s;
    __syncthreads();
    if (grp == 0 && blockIdx.x * 32 + e < n) {
        R t = part[0][e];
#pragma unroll
        for (int q = 1; q < kRedGroups; ++q) t += part[q][e];
        out[k] = t;
    }
}

// loss[b] = full[b] - aligned[b]  (asg.py:128,136) and its reduction (asg.py:137-142), one workgroup,
// fixed-order tree -> deterministic.
template <typename R>
__global__ void __launch_bounds__(256) loss_reduce_kernel(const R *full, const R *aligned, int B, int reduction, R *out) {
    __shared__ double part[256];
    double s = 0;
    for (int b = threadIdx.x; b < B; b += 256) {
        R l = full[b] - aligned[b];
        if (reduction == 0) out[b] = l;
        s += (double) l;
    }
    if (reduction == 0) return;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int) threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (R) (reduction == 2 ? part[0] / B : part[0]);
}

}  // namespace
}  // namespace asg
