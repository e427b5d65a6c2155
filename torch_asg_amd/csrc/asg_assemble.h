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

// ------------------------------------------------------------------ the same assembly on the matrix cores (fp32)
// The per-frame code above spends its time in two N x N VALU products per frame (row sums, outer product) whose operand
// is broadcast through LDS.  Over a BLOCK of 16 frames both are dense contractions:
//   S^T[f][i] = sum_j P[f][j] Ehat[i][j]          (K = labels)   -> row sums of all 16 frames
//   G[i][j]  += sum_f U[f][i] P[f][j]             (K = frames)   -> the outer products, U = posterior / S
// so they go to v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered).  Element layout of a block ("natural"): lane l = 16 g + m
// holds label 16 r + m of frame tb + 4 g + q in register (r, q) -- which is at once the A and B operand layout of the
// second product (k-slot g of k-step q) and the accumulator layout of the first, so the only data movement is ONE
// transpose of P through LDS (the A operand of the first product wants frames along the lanes).  Everything per element
// (posterior, reciprocal, stores) is done with all 64 lanes busy; the per-frame reductions are 16-lane DPP rows.
// Rows whose sum leaves the safe range flag the workgroup, which then redoes its frames with the per-frame code.
// Four independent 16-lane row reductions interleaved (every DPP source is four instructions old: no wait states).
#define ASG_ROW4(op, ctl) \
    op " %0, %0, %0 " ctl " row_mask:0xf bank_mask:0xf\n" op " %1, %1, %1 " ctl " row_mask:0xf bank_mask:0xf\n" \
    op " %2, %2, %2 " ctl " row_mask:0xf bank_mask:0xf\n" op " %3, %3, %3 " ctl " row_mask:0xf bank_mask:0xf\n"
__device__ __forceinline__ void row16_allmax4(float (&x)[4]) {
    asm volatile("s_nop 1\n"
                 ASG_ROW4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") ASG_ROW4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
                 ASG_ROW4("v_max_f32_dpp", "row_half_mirror") ASG_ROW4("v_max_f32_dpp", "row_mirror")
                 "s_nop 1\n"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}
__device__ __forceinline__ void row16_allsum4(float (&x)[4]) {
    asm volatile("s_nop 1\n"
                 ASG_ROW4("v_add_f32_dpp", "quad_perm:[1,0,3,2]") ASG_ROW4("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
                 ASG_ROW4("v_add_f32_dpp", "row_half_mirror") ASG_ROW4("v_add_f32_dpp", "row_mirror")
                 "s_nop 1\n"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}
#undef ASG_ROW4
// Bitwise selects: no control flow (hipcc turns `c ? exp2(x) : 0` into an EXEC-masked branch per element) and NaN-proof
// (frames past an utterance's length hold whatever the allocation held).
__device__ __forceinline__ unsigned bmask(bool c) { return c ? 0xffffffffu : 0u; }
__device__ __forceinline__ float bsel(unsigned mk, float x, float other) {
    return __uint_as_float((__float_as_uint(x) & mk) | (__float_as_uint(other) & ~mk));
}
__device__ __forceinline__ float band(unsigned mk, float x) { return __uint_as_float(__float_as_uint(x) & mk); }

// ---- variant that RECOMPUTES the row sums (round 2's kernel): for batches whose states and emissions do not fit the 256 MB
// memory-side cache.  It reads nothing but the four state arrays (the scale-log variant below reads the emissions again: 22 %
// more traffic, which at B = 4096 costs more than its 30 matrix instructions save: 937 against 890 us per step; at B = 512,
// where everything the forward launch touched is still cache-resident, it is the other way round: 131 against 139 us).
// The aligned lattice's share of the same frames is batched the same way: lane 16 g + m holds target position 16 r + m of
// frame tb + 4 g + q, the per-frame softmax is a 16-lane DPP row reduction, the scatter back to labels goes through a
// per-wavefront fixed-point LDS frame buffer (integer adds commute: deterministic) that is read back in the natural
// layout, so every grad_inputs row is written once, complete.
template <int NP> struct MfmaRsLds {
    static constexpr int NT = (NP + 15) / 16, KS = (NP + 3) / 4, STR = 16 * NT + 4;
    union {
        struct {
            float pt[4][16 * STR];               // per wavefront: P of the block's frames, [frame][label]
            unsigned fxI[4][16][16 * NT];        // per wavefront: aligned state posteriors of the block, scattered to labels
        } blk;
        float tileF[NP <= 48 ? 4 : 2][NP * NP + 1];   // after the frame loop: the wavefronts' tiles (+1: dump slot of padding)
    };
    float eb[NT * KS][64];                       // B operand of the row-sum product (Ehat in k-step order)
    unsigned long long fxT[NP * NP];             // aligned edge posteriors (unscaled, fixed point)
};

template <int NP, int ST>
__global__ void __launch_bounds__(256, (ST <= 2 && NP <= 48) ? 2 : 1) bwd_mfma_rs_kernel(Problem P, State W, BwdArgs A, int parts) {
    typedef float R;
    constexpr int NT = (NP + 15) / 16, KS = (NP + 3) / 4, STR = 16 * NT + 4;
    union Lds {
        AssembleLds<float, NP, 4> S;             // the per-frame code (exact redo of a flagged workgroup)
        MfmaRsLds<NP> M;
    };
    __shared__ Lds L;
    __shared__ int s_bad;
    MfmaRsLds<NP> &M = L.M;
    const int b = blockIdx.x, chunk = blockIdx.y;
    R *tile_out = (R *) A.scratch + ((int64_t) b * A.nchunks + chunk) * P.N * P.N;
    const bool do_ali = (parts & 2) && P.targets;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, m = lane & 15;
    const int N = P.N, T = P.T, S = P.S;
    const R NINF = Num<R>::ninf(), LZ = Num<R>::logzero();
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const int ol = do_ali ? (P.tg_len ? clampi(P.tg_len[b], 0, S) : S) : 0;
    const int t0 = chunk * A.chunk, t1 = min(T, t0 + A.chunk), lim = min(t1, len);
    const R g0 = A.unit_grad ? (R) A.gscale
                             : (A.grad_full ? (R) ((double) ((const R *) A.grad_full)[(int64_t) b * A.gstride] * A.gscale) : R(0));
    const R gf = g0;
    const R ga = (parts & 2) ? (A.neg_aligned ? -g0 : (R) ((double) ((const R *) A.grad_aligned)[(int64_t) b * A.gstride] * A.gscale))
                             : R(0);
    // states through raw buffer loads: lane offset = (clamped frame row) * row bytes + 4 m, the label / position tile is
    // the instruction's immediate offset; elements past a row's end are masked below, past the buffer's end read 0
    __amdgpu_buffer_rsrc_t r_ah = make_rsrc((R *) W.ah + (int64_t) b * T * N, (unsigned) T * (unsigned) N * 4u);
    __amdgpu_buffer_rsrc_t r_bh = make_rsrc((R *) W.bh + (int64_t) b * T * N, (unsigned) T * (unsigned) N * 4u);
    __amdgpu_buffer_rsrc_t r_ab = make_rsrc((R *) W.ab + (int64_t) b * T * S, (unsigned) T * (unsigned) S * 4u);
    __amdgpu_buffer_rsrc_t r_bb = make_rsrc((R *) W.bb + (int64_t) b * T * S, (unsigned) T * (unsigned) S * 4u);
    __amdgpu_buffer_rsrc_t rs_g = make_rsrc((R *) A.grad_inputs + (int64_t) b * N,
                                            (unsigned) ((int64_t) (T - 1) * P.B * N + N) * 4u);
    const unsigned rbN = (unsigned) N * 4u, rbS = (unsigned) S * 4u, rbG = (unsigned) P.B * (unsigned) N * 4u;
    const unsigned m4 = (unsigned) m * 4u;
    // position s - 1 of the previous frame: one element to the left (position 0 has no left neighbour: masked by Dp = logzero)
    const unsigned m4l = m4 >= 4u ? m4 - 4u : 0u;
    struct BlockRegs {
        R a[NT][4], bh[NT][4], ap0[NT];
    };
    struct AlignedRegs {
        R xa[ST][4], xb[ST][4], xm[ST][4], xp0[ST];
    };
    auto issue_aligned = [&](AlignedRegs &X, int tb) {
        const int tf = tb + 4 * g;
        unsigned row[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) row[q] = (unsigned) min(tf + q, T - 1);
        const unsigned rowp = (unsigned) clampi(tf - 1, 0, T - 1);
#pragma unroll
        for (int r = 0; r < ST; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                X.xa[r][q] = buf_load<R>(r_ab, row[q] * rbS + m4 + 64u * r, 0u);
                X.xb[r][q] = buf_load<R>(r_bb, row[q] * rbS + m4 + 64u * r, 0u);
                const unsigned rp = q == 0 ? rowp : row[q - 1];
                X.xm[r][q] = buf_load<R>(r_ab, rp * rbS + (r == 0 ? m4l : m4 + 64u * r - 4u), 0u);
            }
            X.xp0[r] = buf_load<R>(r_ab, rowp * rbS + m4 + 64u * r, 0u);
        }
    };
    auto issue_loads = [&](BlockRegs &X, int tb) {
        const int tf = tb + 4 * g;
        unsigned row[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) row[q] = (unsigned) min(tf + q, T - 1);
        const unsigned rowp = (unsigned) clampi(tf - 1, 0, T - 1);
#pragma unroll
        for (int r = 0; r < NT; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                X.a[r][q] = buf_load<R>(r_ah, row[q] * rbN + m4 + 64u * r, 0u);
                X.bh[r][q] = buf_load<R>(r_bh, row[q] * rbN + m4 + 64u * r, 0u);
            }
            X.ap0[r] = buf_load<R>(r_ah, rowp * rbN + m4 + 64u * r, 0u);
        }
    };
    // the first block's states are requested before anything else: the prologue below (Ehat into LDS, target tables)
    // then runs inside their latency instead of in front of it
    BlockRegs X0;
    if (t0 + 16 * wave < t1) issue_loads(X0, t0 + 16 * wave);
    const R *ehat = (const R *) W.ehat;
    {   // every load in flight before the first LDS write (a rolled loop pays the L2 latency once per trip)
        constexpr int CNT = (NT * KS * 64 + 255) / 256;
        R ev[CNT];
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            const int idx = threadIdx.x + 256 * n;
            const int l = idx & 63, rk = idx >> 6, r = rk / KS, kk = rk - r * KS;
            const int i = 16 * r + (l & 15), j = 4 * kk + (l >> 4);
            const R v = ehat[(int64_t) min(i, N - 1) * W.npad + min(j, N - 1)];
            ev[n] = (i < N && j < N) ? v : R(0);
        }
#pragma unroll
        for (int n = 0; n < CNT; ++n) {
            const int idx = threadIdx.x + 256 * n;
            if (idx < NT * KS * 64) (&M.eb[0][0])[idx] = ev[n];
        }
    }
    for (int k = threadIdx.x; k < N * N; k += 256) M.fxT[k] = 0;
    for (int k = lane; k < 16 * 16 * NT; k += 64) (&M.blk.fxI[wave][0][0])[k] = 0;
    if (threadIdx.x == 0) s_bad = 0;

    bool lv[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) lv[r] = 16 * r + m < N;
    // aligned lattice: this lane's target positions
    bool sv[ST];
    int sc[ST], sm1[ST], tgt[ST], prv[ST];
    R H2[ST], Dp[ST], accH[ST], accD[ST];
#pragma unroll
    for (int r = 0; r < ST; ++r) {
        const int s = 16 * r + m;
        sv[r] = do_ali && s < S;
        sc[r] = sv[r] ? s : 0;
        sm1[r] = (sv[r] && s >= 1) ? s - 1 : 0;
        V2<R> hd = {0, 0};
        int2 tp = {0, 0};
        if (do_ali) {
            hd = reinterpret_cast<const V2<R> *>(W.asu)[(int64_t) b * S + sc[r]];
            tp = reinterpret_cast<const int2 *>(W.asi)[(int64_t) b * S + sc[r]];
        }
        H2[r] = hd.x; Dp[r] = hd.y; tgt[r] = tp.x; prv[r] = tp.y;
        accH[r] = 0; accD[r] = 0;
    }
    V4<R> acc[NT * NT];
#pragma unroll
    for (int q = 0; q < NT * NT; ++q) acc[q] = V4<R>{0, 0, 0, 0};
    bool bad = false;
    float *ptw = M.blk.pt[wave];
    __syncthreads();

    unsigned lvm[NT], lvo[NT], svm[ST], s1m[ST];
#pragma unroll
    for (int r = 0; r < NT; ++r) { lvm[r] = bmask(lv[r]); lvo[r] = lv[r] ? 0u : kOobOffset; }
#pragma unroll
    for (int r = 0; r < ST; ++r) { svm[r] = bmask(sv[r]); s1m[r] = bmask(16 * r + m >= 1); }
    const R LZ2 = LZ + LZ;

    auto process = [&](BlockRegs &C, int tb) {
        const int tf = tb + 4 * g;               // first of this lane group's four frames
        if (tb >= len) {                          // beyond the utterance: zero rows
#pragma unroll
            for (int r = 0; r < NT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    buf_store(R(0), rs_g, (lv[r] && tf + q < t1) ? (unsigned) (tf + q) * rbG + m4 + 64u * r : kOobOffset, 0u);
            return;
        }
        unsigned fvm[4], t1m[4];                  // frame inside the utterance / has a predecessor
#pragma unroll
        for (int q = 0; q < 4; ++q) { fvm[q] = bmask(tf + q < lim); t1m[q] = fvm[q] & bmask(tf + q >= 1); }
        AlignedRegs Q;
        if (do_ali) issue_aligned(Q, tb);       // consumed after the full-lattice part of the block
        // ---- full lattice
        R w[NT][4], p[NT][4];
        R mg[4], Z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            R mx = NINF;
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                const R prev = q == 0 ? C.ap0[r] : C.a[r][q - 1];
                p[r][q] = Num<R>::exp2(bsel(t1m[q] & lvm[r], prev, NINF));
                w[r][q] = bsel(fvm[q] & lvm[r], C.a[r][q] + C.bh[r][q], NINF);
                mx = fmaxf(mx, w[r][q]);
            }
            mg[q] = fmaxf(mx, LZ);
        }
        // the states are consumed: the next block's travel in the same registers while the rest of this one is processed
        // (a second register set instead cost 27 VGPRs and 9 % at B = 4096)
        if (tb + 64 < lim) issue_loads(C, tb + 64);
        row16_allmax4(mg);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            R z = 0;
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                w[r][q] = Num<R>::exp2(w[r][q] - mg[q]);
                z += w[r][q];
            }
            Z[q] = z;
        }
        row16_allsum4(Z);
#pragma unroll
        for (int q = 0; q < 4; ++q) Z[q] = band(bmask(Z[q] > 0), gf * Num<R>::rcp(Z[q]));
        // P transposed through LDS: written [frame][label], read with the frame along the lanes
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) ptw[(4 * g + q) * STR + 16 * r + m] = p[r][q];
        __builtin_amdgcn_wave_barrier();
        V4<R> sd[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r) sd[r] = V4<R>{0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const R pa = ptw[m * STR + 4 * kk + g];
#pragma unroll
            for (int r = 0; r < NT; ++r)
                sd[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, M.eb[r * KS + kk][lane], sd[r], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        R u[NT][4];
        unsigned badm = 0;
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const R post = w[r][q] * Z[q];
                const unsigned sb = Rng<R>::bits(sd[r][q]);
                const unsigned okm = bmask(sb - Rng<R>::lo <= Rng<R>::hi - Rng<R>::lo);      // lo <= sb <= hi
                const unsigned live = t1m[q] & bmask(post != R(0));
                badm |= live & ~okm;
                u[r][q] = band(live & okm, post * Num<R>::rcp(sd[r][q]));
                w[r][q] = post;
            }
        bad |= badm != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ri = 0; ri < NT; ++ri)
#pragma unroll
                for (int rj = 0; rj < NT; ++rj)
                    acc[ri * NT + rj] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[ri][q], p[rj][q], acc[ri * NT + rj], 0, 0, 0);
        // ---- aligned lattice: state posteriors -> label frame buffer, edge posteriors -> accH / accD
        if (do_ali) {
            R gm[ST][4], mg2[4], Z2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R mx = LZ2;
#pragma unroll
                for (int r = 0; r < ST; ++r) {
                    gm[r][q] = bsel(fvm[q] & svm[r], Q.xa[r][q] + Q.xb[r][q], LZ2);
                    mx = fmaxf(mx, gm[r][q]);
                }
                mg2[q] = mx;
            }
            row16_allmax4(mg2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R z = 0;
#pragma unroll
                for (int r = 0; r < ST; ++r) {
                    gm[r][q] = Num<R>::exp2(gm[r][q] - mg2[q]);
                    z += gm[r][q];
                }
                Z2[q] = z;
            }
            row16_allsum4(Z2);
#pragma unroll
            for (int q = 0; q < 4; ++q)      // an infeasible alignment (all states at log zero) has no posterior
                Z2[q] = band(bmask(mg2[q] > R(-1e29)) & bmask(Z2[q] > 0), Num<R>::rcp(Z2[q]));
#pragma unroll
            for (int r = 0; r < ST; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const R post2 = gm[r][q] * Z2[q];
                    atomicAdd(&M.blk.fxI[wave][4 * g + q][tgt[r]], FrameFix<R>::to(post2));
                    // stay / arrive shares of the state posterior: softmax over the two incoming edges,
                    // 1 / (1 + 2^-|d|) and its complement (one exp2 and one rcp instead of a log-sum-exp and two exp2)
                    const R pc0 = (q == 0 ? Q.xp0[r] : Q.xa[r][q - 1]) + H2[r];
                    const R pc1 = band(s1m[r], Q.xm[r][q]) + Dp[r];
                    const R d = pc1 - pc0;
                    const R tt = Num<R>::exp2(-fabsf(d));
                    const R big = Num<R>::rcp(R(1) + tt), small = tt * big;
                    const unsigned em = t1m[q] & bmask(post2 != R(0));
                    accH[r] += band(em, post2 * (d <= R(0) ? big : small));
                    accD[r] += band(em, post2 * (d <= R(0) ? small : big));
                }
        }
        // ---- rows: full posterior + the aligned posteriors scattered to this label
        unsigned so[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) so[q] = tf + q < t1 ? (unsigned) (tf + q) * rbG + m4 : kOobOffset;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R v = w[r][q];
                if (do_ali) {
                    unsigned *fp = &M.blk.fxI[wave][4 * g + q][16 * r + m];
                    v += ga * FrameFix<R>::from(*fp);
                    *fp = 0;
                }
                buf_store(v, rs_g, max(so[q] + 64u * r, lvo[r]), 0u);
            }
        __builtin_amdgcn_wave_barrier();
    };

    for (int tb = t0 + 16 * wave; tb < t1; tb += 64) process(X0, tb);
    if (do_ali) {
#pragma unroll
        for (int r = 0; r < ST; ++r) {
            const int s = 16 * r + m;
            if (s < ol) {
                if (accH[r] != R(0)) atomicAdd(&M.fxT[tgt[r] * N + tgt[r]], to_fix<R>(accH[r]));
                if (s >= 1 && accD[r] != R(0)) atomicAdd(&M.fxT[tgt[r] * N + prv[r]], to_fix<R>(accD[r]));
            }
        }
    }
    if (__any(bad) && lane == 0) s_bad = 1;
    __syncthreads();
    if (s_bad) {         // rare: the per-frame code owns the exact treatment of unusable row sums
        assemble_frames<float, NP, 4>(P, W, A, parts, b, chunk, tile_out, L.S);
        return;
    }
    // one partial tile per workgroup: the four wavefronts' accumulators in a fixed order, scaled by Ehat
    constexpr int CT = (NP * NP + 255) / 256;
    R eh[CT];
#pragma unroll
    for (int n = 0; n < CT; ++n) {           // in flight during the combine below
        const int k = min((int) threadIdx.x + 256 * n, N * N - 1), i = k / N, j = k - i * N;
        eh[n] = ehat[(int64_t) i * W.npad + j];
    }
    // every wavefront stores its accumulators to its own slot (plain pipelined LDS writes; a read-modify-write per
    // element pays the LDS latency per element); alphabets too large for four slots take two rounds
    constexpr int TW = NP <= 48 ? 4 : 2;
    for (int ph = 0; ph < 4 / TW; ++ph) {
        if (wave / TW == ph) {
            float *slot = M.tileF[wave % TW];
#pragma unroll
            for (int ri = 0; ri < NT; ++ri)
#pragma unroll
                for (int rj = 0; rj < NT; ++rj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 16 * ri + 4 * g + q, j = 16 * rj + m;
                        const int at = (i < N && j < N) ? i * NP + j : NP * NP;
                        slot[at] = (ph == 0 ? 0.f : slot[at]) + acc[ri * NT + rj][q];
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int n = 0; n < CT; ++n) {
        const int k = threadIdx.x + 256 * n;
        if (k < N * N) {
            const int i = k / N, j = k - i * N;
            R t = M.tileF[0][i * NP + j];
#pragma unroll
            for (int w2 = 1; w2 < TW; ++w2) t += M.tileF[w2][i * NP + j];      // fixed order
            R v = t * eh[n];
            const unsigned long long fv = M.fxT[k];
            if (fv != 0) v += ga * from_fix<R>(fv);
            tile_out[k] = v;
        }
    }
}

// The aligned lattice's share of the same frames is batched the same way: lane 16 g + m holds target position 16 r + m of
// frame tb + 4 g + q, the per-frame softmax is a 16-lane DPP row reduction, the scatter back to labels goes through a
// per-wavefront fixed-point LDS frame buffer (integer adds commute: deterministic) that is read back in the natural
// layout, so every grad_inputs row is written once, complete.
//
// Round 4: the row sums are not recomputed.  u_i = posterior_i / s_i with s_i = 2^(ah[t][i] - arg_i), arg_i the exponent of the
// emission factor the alpha pass used (its ScaleLog, asg_kernels.h, gives the two frame scalars; the emission is read again):
//     u_i = g * 2^(bh[t][i] + arg_i - max_t) / Z_t            -- ah[t][i] cancels
// so the first product (30 of the 66 matrix instructions at N = 40, which on gfx950 run on the same vector ALU as everything
// else), its LDS transpose and the per-element range test are gone; a frame the alpha pass produced with its exact per-node
// code has a NaN in the log and sends the workgroup to the per-frame code.  No per-element selects: frames outside the
// utterance load a valid frame's states instead (clamped rows) and drop out through their per-frame factor.
#ifndef ASG_BWD_ABL
#define ASG_BWD_ABL 0           // developer timing probes (wrong results): 1 no row stores, 2 no state / emission loads, 4 no matrix
#endif                          // instructions, 8 no LDS scatter of the aligned posteriors
template <int NP> struct MfmaLds {
    static constexpr int NT = (NP + 15) / 16;
    union {
        unsigned fxI[4][16][16 * NT];            // per wavefront: aligned state posteriors of the block, scattered to labels
        float tileF[NP <= 48 ? 4 : 2][NP * NP + 1];   // after the frame loop: the wavefronts' tiles (+1: dump slot of padding)
    };
    unsigned long long fxT[NP * NP];             // aligned edge posteriors (unscaled, fixed point)
};

template <int NP, int ST, bool ALI>
__global__ void __launch_bounds__(256, (ST <= 2 && NP <= 48) ? 2 : 1) bwd_mfma_kernel(Problem P, State W, BwdArgs A, int parts) {
    typedef float R;
    constexpr int NT = (NP + 15) / 16;
    union Lds {
        AssembleLds<float, NP, 4> S;             // the per-frame code (exact redo of a flagged workgroup)
        MfmaLds<NP> M;
    };
    __shared__ Lds L;
    __shared__ int s_bad;
    MfmaLds<NP> &M = L.M;
    const int b = blockIdx.x, chunk = blockIdx.y;
    R *tile_out = (R *) A.scratch + ((int64_t) b * A.nchunks + chunk) * P.N * P.N;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // which quarter of the chunk's blocks this wavefront takes: rotated with the workgroup's position in the launch, so that the
    // wavefronts that get one block more than the others (25 blocks over 4 wavefronts at T = 400) do not all sit on SIMD 0
    const int wq = (wave + ((blockIdx.x + gridDim.x * blockIdx.y) >> 8)) & 3;
    const int g = lane >> 4, m = lane & 15;
    const int N = P.N, T = P.T, S = P.S;
    const R NINF = Num<R>::ninf(), LZ = Num<R>::logzero(), L2E = Num<R>::log2e();
    const int len = P.in_len ? clampi(P.in_len[b], 0, T) : T;
    const int ol = ALI ? (P.tg_len ? clampi(P.tg_len[b], 0, S) : S) : 0;
    const int t0 = chunk * A.chunk, t1 = min(T, t0 + A.chunk), lim = min(t1, len);
    const R g0 = A.unit_grad ? (R) A.gscale
                             : (A.grad_full ? (R) ((double) ((const R *) A.grad_full)[(int64_t) b * A.gstride] * A.gscale) : R(0));
    const R gf = g0;
    const R ga = ALI ? (A.neg_aligned ? -g0 : (R) ((double) ((const R *) A.grad_aligned)[(int64_t) b * A.gstride] * A.gscale))
                     : R(0);
    // states through raw buffer loads: lane offset = (clamped frame row) * row bytes + 4 m, the label / position tile is
    // the instruction's immediate offset; elements past a row's end are masked below, past the buffer's end read 0
    __amdgpu_buffer_rsrc_t r_ah = make_rsrc((R *) W.ah + (int64_t) b * T * N, (unsigned) T * (unsigned) N * 4u);
    __amdgpu_buffer_rsrc_t r_bh = make_rsrc((R *) W.bh + (int64_t) b * T * N, (unsigned) T * (unsigned) N * 4u);
    __amdgpu_buffer_rsrc_t r_ab = make_rsrc((R *) W.ab + (int64_t) b * T * S, (unsigned) T * (unsigned) S * 4u);
    __amdgpu_buffer_rsrc_t r_bb = make_rsrc((R *) W.bb + (int64_t) b * T * S, (unsigned) T * (unsigned) S * 4u);
    __amdgpu_buffer_rsrc_t r_k = make_rsrc((R *) W.klog + (int64_t) b * T * 2, (unsigned) T * 8u);
    // emissions: frame and label offsets in elements of the caller's strides (launch_bwd_small checks that they fit 32 bits)
    __amdgpu_buffer_rsrc_t r_in = make_rsrc((R *) P.inputs + (int64_t) b * P.is1, 0xffffffffu);
    __amdgpu_buffer_rsrc_t rs_g = make_rsrc((R *) A.grad_inputs + (int64_t) b * N,
                                            (unsigned) ((int64_t) (T - 1) * P.B * N + N) * 4u);
    const unsigned rbN = (unsigned) N * 4u, rbS = (unsigned) S * 4u, rbG = (unsigned) P.B * (unsigned) N * 4u;
    const unsigned rbX = (unsigned) P.is0 * 4u, tileX = 16u * (unsigned) P.is2 * 4u;
    // (labels past N in the last tile read label N - 1: inside the tensor whatever its strides; their values only reach tile rows >= N)
    const unsigned m4 = (unsigned) m * 4u, mX = (unsigned) m * (unsigned) P.is2 * 4u;
    const unsigned mXl = (unsigned) min(16 * (NT - 1) + m, N - 1) * (unsigned) P.is2 * 4u;
    // position s - 1 of the previous frame: one element to the left (position 0 has no left neighbour: masked by Dp = logzero)
    const unsigned m4l = m4 >= 4u ? m4 - 4u : 0u;
    struct BlockRegs {
        R a[NT][4], bh[NT][4], ap0[NT], x[NT][4];
        V2<R> k[4];                              // scale log of the lane group's four frames: {zb, ex}
    };
    struct AlignedRegs {
        R xa[ST][4], xb[ST][4], xm[ST][4], xp0[ST];
    };
    // Frame rows are clamped into the utterance's part of the chunk, [0, lim): a frame outside it reads a valid frame's states
    // (finite, so nothing below needs a per-element select) and is taken out by its per-frame factor (Zw / Zu / Z2 below).
    const int rmaxf = max(lim, 1) - 1;
    auto issue_aligned = [&](AlignedRegs &X, int tb) {
        if (ASG_BWD_ABL & 2) {
#pragma unroll
            for (int r = 0; r < ST; ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { asm volatile("" : "=v"(X.xa[r][q])); asm volatile("" : "=v"(X.xb[r][q])); asm volatile("" : "=v"(X.xm[r][q])); }
                asm volatile("" : "=v"(X.xp0[r]));
            }
            return;
        }
        const int tf = tb + 4 * g;
        unsigned row[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) row[q] = (unsigned) min(tf + q, rmaxf);
        const unsigned rowp = (unsigned) clampi(tf - 1, 0, rmaxf);
#pragma unroll
        for (int r = 0; r < ST; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                X.xa[r][q] = buf_load<R>(r_ab, row[q] * rbS + m4 + 64u * r, 0u);
                X.xb[r][q] = buf_load<R>(r_bb, row[q] * rbS + m4 + 64u * r, 0u);
                const unsigned rp = q == 0 ? rowp : row[q - 1];
                X.xm[r][q] = buf_load<R>(r_ab, rp * rbS + (r == 0 ? m4l : m4 + 64u * r - 4u), 0u);
            }
            X.xp0[r] = buf_load<R>(r_ab, rowp * rbS + m4 + 64u * r, 0u);
        }
    };
    auto issue_loads = [&](BlockRegs &X, int tb) {
        if (ASG_BWD_ABL & 2) {
#pragma unroll
            for (int r = 0; r < NT; ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { asm volatile("" : "=v"(X.a[r][q])); asm volatile("" : "=v"(X.bh[r][q])); asm volatile("" : "=v"(X.x[r][q])); }
                asm volatile("" : "=v"(X.ap0[r]));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) X.k[q] = V2<R>{R(1), R(0)};
            return;
        }
        const int tf = tb + 4 * g;
        unsigned row[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) row[q] = (unsigned) min(tf + q, rmaxf);
        const unsigned rowp = (unsigned) clampi(tf - 1, 0, rmaxf);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            const u2v kk = __builtin_amdgcn_raw_buffer_load_b64(r_k, row[q] * 8u, 0u, 0);
            X.k[q] = V2<R>{__uint_as_float(kk.x), __uint_as_float(kk.y)};
        }
#pragma unroll
        for (int r = 0; r < NT; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                X.a[r][q] = buf_load<R>(r_ah, row[q] * rbN + m4 + 64u * r, 0u);
                X.bh[r][q] = buf_load<R>(r_bh, row[q] * rbN + m4 + 64u * r, 0u);
                X.x[r][q] = r == NT - 1 ? buf_load<R>(r_in, row[q] * rbX + mXl, 0u) : buf_load<R>(r_in, row[q] * rbX + mX, tileX * (unsigned) r);
            }
            X.ap0[r] = buf_load<R>(r_ah, rowp * rbN + m4 + 64u * r, 0u);
        }
    };
    // the first block's states are requested before anything else: the prologue below then runs inside their latency
    BlockRegs X0;
    if (t0 + 16 * wq < lim) issue_loads(X0, t0 + 16 * wq);
    const R *ehat = (const R *) W.ehat;
    const R mark = ((const R *) W.klog)[(int64_t) b * T * 2];
    R Ri[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) Ri[r] = ((const R *) W.rmax)[min(16 * r + m, N - 1)];
    for (int k = threadIdx.x; k < N * N; k += 256) M.fxT[k] = 0;
    if (ALI)
        for (int k = lane; k < 16 * 16 * NT; k += 64) (&M.fxI[wave][0][0])[k] = 0;
    if (threadIdx.x == 0) s_bad = 0;

    const bool lv_last = 16 * (NT - 1) + m < N;          // (tiles before the last are complete: NP - 8 < N)
    // aligned lattice: this lane's target positions
    bool sv[ST];
    int tgt[ST], prv[ST];
    R H2[ST], Dp[ST], accH[ST], accD[ST];
#pragma unroll
    for (int r = 0; r < ST; ++r) {
        const int s = 16 * r + m;
        sv[r] = ALI && s < S;
        const int sc = sv[r] ? s : 0;
        V2<R> hd = {0, 0};
        int2 tp = {0, 0};
        if (ALI) {
            hd = reinterpret_cast<const V2<R> *>(W.asu)[(int64_t) b * S + sc];
            tp = reinterpret_cast<const int2 *>(W.asi)[(int64_t) b * S + sc];
        }
        H2[r] = hd.x; Dp[r] = hd.y; tgt[r] = tp.x; prv[r] = tp.y;
        accH[r] = 0; accD[r] = 0;
    }
    V4<R> acc[NT * NT];
#pragma unroll
    for (int q = 0; q < NT * NT; ++q) acc[q] = V4<R>{0, 0, 0, 0};
    // an alpha pass that left no scale log (none does: a stale buffer would say so) -> the per-frame code
    bool bad = len >= 1 && !(mark == R(kScaleLogMark));
    __syncthreads();

    const unsigned lvm_last = bmask(lv_last), lvo_last = lv_last ? 0u : kOobOffset;
    unsigned svm[ST];
#pragma unroll
    for (int r = 0; r < ST; ++r) svm[r] = bmask(sv[r]);
    const R LZ2 = LZ + LZ;
    const R gafx = ga * (1.0f / 1073741824.0f);          // aligned gradient weight times the fixed-point quantum of the label scatter

    auto process = [&](BlockRegs &C, int tb) {
        const int tf = tb + 4 * g;               // first of this lane group's four frames
        if (tb >= lim) {                          // beyond the utterance: zero rows
#pragma unroll
            for (int r = 0; r < NT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    buf_store(R(0), rs_g, ((r < NT - 1 || lv_last) && tf + q < t1) ? (unsigned) (tf + q) * rbG + m4 + 64u * r : kOobOffset, 0u);
            return;
        }
        unsigned fvm[4], t1m[4];                  // frame inside the utterance / has a predecessor
#pragma unroll
        for (int q = 0; q < 4; ++q) { fvm[q] = bmask(tf + q < lim); t1m[q] = fvm[q] & bmask(tf + q >= 1); }
        AlignedRegs Q;
        if (ALI) issue_aligned(Q, tb);          // consumed after the full-lattice part of the block
        // ---- full lattice
        R w[NT][4], p[NT][4], ua[NT][4];
        R mg[4], Z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // a frame without an incoming edge (frame 0 -- whose log entry is the validity mark -- or a clamped repeat of the last
            // frame) gets an emission exponent of -inf: its u is then exactly 0, whatever its emissions are (u = 2^(..) * 0 with
            // an overflowing power would be NaN: frame 0's state is not scaled like the others')
            const R kz = bsel(t1m[q], C.k[q].x, __builtin_inff()), kx = C.k[q].y;
            bad |= !(kz == kz);                  // NaN: a frame of the exact per-node code
            R mx = NINF;
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                // (labels past N: whatever the neighbouring row holds; they only reach tile rows / columns >= N)
                p[r][q] = Num<R>::exp2(q == 0 ? C.ap0[r] : C.a[r][q - 1]);
                R ww = C.a[r][q] + C.bh[r][q];
                if (r == NT - 1) ww = bsel(lvm_last, ww, NINF);
                w[r][q] = ww;
                mx = fmaxf(mx, ww);
                // exponent of the alpha pass's emission factor, rounded as the alpha pass rounded it
                ua[r][q] = C.bh[r][q] + (__builtin_fmaf(C.x[r][q], L2E, Ri[r] - kz) - kx);
            }
            mg[q] = fmaxf(mx, LZ);
        }
        // the states are consumed: the next block's travel in the same registers while the rest of this one is processed
        // (a second register set instead cost 27 VGPRs and 9 % at B = 4096)
        __builtin_amdgcn_sched_barrier(0);
        if (tb + 64 < lim) issue_loads(C, tb + 64);
        __builtin_amdgcn_sched_barrier(0);
        row16_allmax4(mg);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            R z = 0;
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                w[r][q] = Num<R>::exp2(w[r][q] - mg[q]);
                z += w[r][q];
            }
            Z[q] = z;
        }
        row16_allsum4(Z);
        R Zw[4], Zu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const R zi = band(bmask(Z[q] > 0), gf * Num<R>::rcp(Z[q]));
            Zw[q] = band(fvm[q], zi);            // frames outside the utterance: zero row, no edge
            Zu[q] = band(t1m[q], zi);            // frame 0: no edge
        }
        R u[NT][4];
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u[r][q] = Num<R>::exp2(ua[r][q] - mg[q]) * Zu[q];           // posterior / row sum
                w[r][q] *= Zw[q];                                            // posterior (times the upstream gradient)
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ri = 0; ri < NT; ++ri)
#pragma unroll
                for (int rj = 0; rj < NT; ++rj)
                    if (ASG_BWD_ABL & 4) acc[ri * NT + rj][q] += u[ri][q] * p[rj][q];
                    else acc[ri * NT + rj] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[ri][q], p[rj][q], acc[ri * NT + rj], 0, 0, 0);
        // ---- aligned lattice: state posteriors -> label frame buffer, edge posteriors -> accH / accD
        __builtin_amdgcn_sched_barrier(0);
        if (ALI) {
            R gm[ST][4], mg2[4], Z2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R mx = LZ2;
#pragma unroll
                for (int r = 0; r < ST; ++r) {
                    gm[r][q] = bsel(svm[r], Q.xa[r][q] + Q.xb[r][q], LZ2);
                    mx = fmaxf(mx, gm[r][q]);
                }
                mg2[q] = mx;
            }
            row16_allmax4(mg2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R z = 0;
#pragma unroll
                for (int r = 0; r < ST; ++r) {
                    gm[r][q] = Num<R>::exp2(gm[r][q] - mg2[q]);
                    z += gm[r][q];
                }
                Z2[q] = z;
            }
            row16_allsum4(Z2);
            R Z2e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {    // an infeasible alignment (all states at log zero) has no posterior
                const R zi = band(bmask(mg2[q] > R(-1e29)) & bmask(Z2[q] > 0), Num<R>::rcp(Z2[q]));
                Z2[q] = band(fvm[q], zi) * R(1073741824.0);      // (times the fixed-point scale of the label scatter: FrameFix<float>)
                Z2e[q] = band(t1m[q], zi);
            }
#pragma unroll
            for (int r = 0; r < ST; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!(ASG_BWD_ABL & 8)) atomicAdd(&M.fxI[wave][4 * g + q][tgt[r]], (unsigned) __builtin_rintf(gm[r][q] * Z2[q]));
                    // stay / arrive shares of the state posterior: softmax over the two incoming edges,
                    // 1 / (1 + 2^-|d|) and its complement (one exp2 and one rcp instead of a log-sum-exp and two exp2).
                    // Position 0 has no left neighbour: its arrive share dies with Dp = log zero (the state read for it is
                    // position 0's own, finite).
                    const R post2 = gm[r][q] * Z2e[q];
                    const R pc0 = (q == 0 ? Q.xp0[r] : Q.xa[r][q - 1]) + H2[r];
                    const R pc1 = Q.xm[r][q] + Dp[r];
                    const R d = pc1 - pc0;
                    const R tt = Num<R>::exp2(-fabsf(d));
                    const R big = Num<R>::rcp(R(1) + tt), small = tt * big;
                    accH[r] += post2 * (d <= R(0) ? big : small);
                    accD[r] += post2 * (d <= R(0) ? small : big);
                }
        }
        // ---- rows: full posterior + the aligned posteriors scattered to this label
        __builtin_amdgcn_sched_barrier(0);
        unsigned so[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) so[q] = tf + q < t1 ? (unsigned) (tf + q) * rbG + m4 : kOobOffset;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R v = w[r][q];
                if (ALI) {
                    unsigned *fp = &M.fxI[wave][4 * g + q][16 * r + m];
                    v = __builtin_fmaf((float) *fp, gafx, v);
                    *fp = 0;
                }
                if (ASG_BWD_ABL & 1) { if (v == R(12345.678f)) buf_store(v, rs_g, so[q], 0u); }
                else buf_store(v, rs_g, r == NT - 1 ? max(so[q] + 64u * r, lvo_last) : so[q] + 64u * r, 0u);
            }
        __builtin_amdgcn_wave_barrier();
    };

    for (int tb = t0 + 16 * wq; tb < t1; tb += 64) process(X0, tb);
    if (ALI) {
#pragma unroll
        for (int r = 0; r < ST; ++r) {
            const int s = 16 * r + m;
            if (s < ol) {
                if (accH[r] != R(0)) atomicAdd(&M.fxT[tgt[r] * N + tgt[r]], to_fix<R>(accH[r]));
                if (s >= 1 && accD[r] != R(0)) atomicAdd(&M.fxT[tgt[r] * N + prv[r]], to_fix<R>(accD[r]));
            }
        }
    }
    if (__any(bad) && lane == 0) s_bad = 1;
    __syncthreads();
    if (s_bad) {         // rare: the per-frame code owns the exact treatment of unusable row sums
        assemble_frames<float, NP, 4>(P, W, A, parts, b, chunk, tile_out, L.S);
        return;
    }
    // one partial tile per workgroup: the four wavefronts' accumulators in a fixed order, scaled by Ehat
    constexpr int CT = (NP * NP + 255) / 256;
    R eh[CT];
#pragma unroll
    for (int n = 0; n < CT; ++n) {           // in flight during the combine below
        const int k = min((int) threadIdx.x + 256 * n, N * N - 1), i = k / N, j = k - i * N;
        eh[n] = ehat[(int64_t) i * W.npad + j];
    }
    // every wavefront stores its accumulators to its own slot (plain pipelined LDS writes; a read-modify-write per
    // element pays the LDS latency per element); alphabets too large for four slots take two rounds
    constexpr int TW = NP <= 48 ? 4 : 2;
    for (int ph = 0; ph < 4 / TW; ++ph) {
        if (wave / TW == ph) {
            float *slot = M.tileF[wave % TW];
#pragma unroll
            for (int ri = 0; ri < NT; ++ri)
#pragma unroll
                for (int rj = 0; rj < NT; ++rj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 16 * ri + 4 * g + q, j = 16 * rj + m;
                        const int at = (i < N && j < N) ? i * NP + j : NP * NP;
                        slot[at] = (ph == 0 ? 0.f : slot[at]) + acc[ri * NT + rj][q];
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int n = 0; n < CT; ++n) {
        const int k = threadIdx.x + 256 * n;
        if (k < N * N) {
            const int i = k / N, j = k - i * N;
            R t = M.tileF[0][i * NP + j];
#pragma unroll
            for (int w2 = 1; w2 < TW; ++w2) t += M.tileF[w2][i * NP + j];      // fixed order
            R v = t * eh[n];
            const unsigned long long fv = M.fxT[k];
            if (fv != 0) v += ga * from_fix<R>(fv);
            tile_out[k] = v;
        }
    }
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
