// torch_asg_amd/csrc/asg_generic_grad.hip -- generic path, FULL LATTICE GRADIENT (N > 64): posterior rows from the stored states and the
// logged normalisers, grad_transition = E o (U^T P) as a contraction over the frame axis (fp32 matrix instruction; beyond 1024 labels every
// float as three bfloat16 on v_mfma_f32_32x32x16_bf16), exact fix-up of rows outside the fp32-safe range; and the layout of the backward
// scratch buffer.  Replaces fully_connected_lattice.cpp:49-63 without path_contrib.  The file-level story is in asg_generic.hip.
#include "asg_generic_common.h"

namespace asg {

namespace {

// ------------------------------------------------------------------ gradient: full lattice
// per (b,t) posterior + exp-domain previous frame.  grid = (T, B), block = 256.
//   grad_inputs[t][b][:] = g_b * softmax(ah+bh)      (zeros for t >= len; the aligned part is added later)
//   Pm[(b,t)][:] = exp2(ah[t-1] - max)  (t>=1, else 0)     Gm[(b,t)][:] = g_b * softmax   (t>=1 rows used)
// DIRECT (fp32): Gm receives U = g * softmax / (row sum) at once, WITHOUT the row-sum product.  The forward pass stored
//   ah[t][i] = x2[t][i] - emax[t] + hmax[i] + log2(sum_j Ehat[i][j] exp2(ah[t-1][j])) - mu[t]
// so the row sum against Pm = exp2(ah[t-1] - mp) is  exp2(lambda),  lambda = ah[t][i] - x2[t][i] + emax[t] + mu[t] - hmax[i] - mp:
// five loads and an exp2 per element instead of a [N x N] x [N x BT] contraction (102 ms of cfg 5's 507).  Rows whose
// sum is outside 2^+-100 are marked for bwd_fix_kernel exactly as the contraction's epilogue marked them.
// rowoff (fp32 route): the rows of Pm / Gm are COMPACTED -- only frames 1 .. len-1 of every utterance carry a
// transition (frame 0 has no predecessor, frames >= len are padding: both would be rows of zeros in the contraction over
// the frame axis), row of (b, t) = rowoff[b] + t - 1, rowoff[B] = number of rows = the contraction's K (rowoff_kernel).
template <typename R, bool DIRECT>
__global__ void __launch_bounds__(256) bwd_post_kernel(Problem P, State W, BwdArgs A, R *Pm, R *Gm, int npad, const R *emax,
                                                       const R *mulog, int *anybad, const int *rowoff) {
    __shared__ R red[4];
    const int t = blockIdx.x, b = blockIdx.y, N = P.N, T = P.T;
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const R LZ = Num<R>::logzero();
    R *gin = (R *) A.grad_inputs + ((int64_t) t * P.B + b) * N;
    const int64_t row = rowoff ? (int64_t) rowoff[b] + t - 1 : (int64_t) b * T + t;
    const bool has_row = !rowoff || (t >= 1 && t < len);
    R *pm = Pm + row * npad, *gm = Gm + row * npad;
    if (t >= len) {
        for (int i = threadIdx.x; i < N; i += 256) gin[i] = 0;
        if (has_row) for (int i = threadIdx.x; i < npad; i += 256) { pm[i] = 0; gm[i] = 0; }
        return;
    }
    const R gf = (R) ((double) ((const R *) A.grad_full)[(int64_t) b * A.gstride] * A.gscale);
    const R *ah = (const R *) W.ah + ((int64_t) b * T + t) * N;
    const R *bh = (const R *) W.bh + ((int64_t) b * T + t) * N;
    R m = Num<R>::ninf();
    for (int i = threadIdx.x; i < N; i += 256) m = fmax(m, ah[i] + bh[i]);
    m = fmax(block_reduce_max<R>(m, red), LZ);
    R z = 0;
    for (int i = threadIdx.x; i < N; i += 256) z += Num<R>::exp2(ah[i] + bh[i] - m);
    z = block_reduce_sum<R>(z, red);
    R mp = Num<R>::ninf();
    if (t >= 1) {
        for (int i = threadIdx.x; i < N; i += 256) mp = fmax(mp, ah[i - N]);
        mp = fmax(block_reduce_max<R>(mp, red), LZ);
    }
    for (int i = threadIdx.x; i < npad; i += 256) {
        R g = 0, pv = 0;
        if (i < N) {
            g = (z > 0) ? gf * Num<R>::exp2(ah[i] + bh[i] - m) / z : R(0);
            gin[i] = g;
            if (t >= 1) pv = Num<R>::exp2(ah[i - N] - mp);
        }
        if (!has_row) continue;
        pm[i] = pv;
        if (DIRECT) {
            R u = 0;
            if (t >= 1 && i < N && g != R(0)) {
                const R x2 = ((const R *) P.inputs)[(int64_t) t * P.is0 + (int64_t) b * P.is1 + (int64_t) i * P.is2] * Num<R>::log2e();
                const R lam = ah[i] - x2 + emax[(int64_t) t * P.B + b] + mulog[(int64_t) t * P.B + b] - ((const R *) W.rmax)[i] - mp;
                const bool ok = fabs(lam) < Num<R>::lg_limit();
                u = ok ? g * Num<R>::exp2(-lam) : Num<R>::ninf();
                if (!ok) *anybad = 1;
            }
            gm[i] = u;
        } else {
            gm[i] = (t >= 1) ? g : R(0);
        }
    }
}

// rowoff[b] = sum over b' < b of max(len_b' - 1, 0), rowoff[B] = the total.  One wavefront, utterances in chunks of 64
// (wave prefix sums by DPP-free shuffles: B is small next to T * N).
__global__ void __launch_bounds__(64) rowoff_kernel(Problem P, int *rowoff) {
    const int lane = threadIdx.x, T = P.T, B = P.B;
    int base = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        const int len = b < B ? (P.in_len ? gclampi(P.in_len[b], 0, T) : T) : 0;
        int v = len > 1 ? len - 1 : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (b < B) rowoff[b] = base + incl - v;
        base += __shfl(incl, 63);
    }
    if (lane == 0) rowoff[B] = base;
}

// LDS-tiled product C[m][n] = sum_k A(m,k) * B(k,n), 64x64 tile, 4x4 per thread, BK = 16.
//   MODE 0 (row sums):   A = ehat[m][k] (k contiguous), B(k,n) = Pm[n][k] (k contiguous), epilogue
//                        Gm[n][m] <- (ok) ? Gm[n][m] / C : 0          (U overwrites G in place)
//   MODE 1 (outer prod): A(m,k) = Gm[k][m] (m contiguous), B(k,n) = Pm[k][n] (n contiguous), epilogue
//                        out[m][n] = C * ehat[m][n]
// MODE 1 with kslice > 0: blockIdx.z takes rows [z kslice, (z + 1) kslice) of the frame axis and writes its partial sums to
// partial[z] (no E factor: gemm_combine_kernel adds the slices in order and applies it) -- an output of a few 64 x 64 tiles
// otherwise leaves the device to a handful of workgroups (fp64, N = 128: 4 workgroups, 5.2 ms for a 25 600-row contraction).
template <typename R, int MODE>
__global__ void __launch_bounds__(256) bwd_gemm_kernel(const R *ehat, const R *Pm, R *Gm, R *out, int N, int npad, int K,
                                                       int *anybad, int kslice = 0, R *partial = nullptr) {
    constexpr int BK = 16;
    __shared__ __attribute__((aligned(16))) R As[BK][64 + 4];
    __shared__ __attribute__((aligned(16))) R Bs[BK][64 + 4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int Mdim = N, Ndim = MODE == 0 ? K : N;
    const int kbeg = (MODE == 1 && kslice > 0) ? (int) blockIdx.z * kslice : 0;
    const int Kdim = MODE == 0 ? npad : ((MODE == 1 && kslice > 0) ? min(K, kbeg + kslice) : K);
    R acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0;
    for (int k0 = kbeg; k0 < Kdim; k0 += BK) {
        for (int e = threadIdx.x; e < 64 * BK; e += 256) {
            if (MODE == 0) {
                int mm = e / BK, kk = e - mm * BK;
                int gm_ = m0 + mm, gk = k0 + kk;
                As[kk][mm] = (gm_ < Mdim && gk < Kdim) ? ehat[(int64_t) gm_ * npad + gk] : R(0);
                int nn = mm;
                int gn = n0 + nn;
                Bs[kk][nn] = (gn < Ndim && gk < Kdim) ? Pm[(int64_t) gn * npad + gk] : R(0);
            } else {
                int kk = e / 64, mm = e - kk * 64;
                int gk = k0 + kk, gm_ = m0 + mm, gn = n0 + mm;
                R gv = (gk < Kdim && gm_ < Mdim) ? Gm[(int64_t) gk * npad + gm_] : R(0);
                As[kk][mm] = (gv == Num<R>::ninf()) ? R(0) : gv;
                Bs[kk][mm] = (gk < Kdim && gn < Ndim) ? Pm[(int64_t) gk * npad + gn] : R(0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            R av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[kk][ty * 4 + a];
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = Bs[kk][tx * 4 + c];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = fma(av[a], bv[c], acc[a][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int gm_ = m0 + ty * 4 + a, gn = n0 + tx * 4 + c;
            if (gm_ >= Mdim || gn >= Ndim) continue;
            if (MODE == 0) {
                R g = Gm[(int64_t) gn * npad + gm_];
                R sden = acc[a][c];
                bool ok = fabs(Num<R>::log2(sden)) < Num<R>::lg_limit();
                // rows outside the safe range are marked -inf for the exact fix-up kernel (which recomputes
                // their posterior); the outer-product pass reads markers as 0
                const bool mark = !ok && g != R(0);
                Gm[(int64_t) gn * npad + gm_] = ok ? g / sden : (mark ? Num<R>::ninf() : R(0));
                if (mark) *anybad = 1;
            } else if (kslice > 0) {
                partial[(int64_t) blockIdx.z * N * N + (int64_t) gm_ * N + gn] = acc[a][c];
            } else {
                out[(int64_t) gm_ * N + gn] = acc[a][c] * ehat[(int64_t) gm_ * npad + gn];
            }
        }
}

// The same two products on the matrix cores (fp32 only): 128 x 128 tile per workgroup, 64 x 64 per wavefront (4 x 4
// blocks of v_mfma_f32_16x16x4_f32: exact fp32), BK = 16 staged global -> registers -> LDS with the next tile's loads in
// flight.  Operands come out of LDS in the MFMA's own order: A[m = l & 15][k = l >> 4] = As[k][m], B likewise.
#ifndef ASG_X_GEMM_BK
#define ASG_X_GEMM_BK 32
#endif
template <int MODE>
__global__ void __launch_bounds__(256) bwd_gemm_mfma(const float *ehat, const float *Pm, float *Gm, float *out, int N, int npad,
                                                     int K, int *anybad, const int *kdev, int kslice, float *partial) {
    typedef float R;
    if (kdev) K = __builtin_amdgcn_readfirstlane(*kdev);        // compacted rows: their number is known on the device only
    // MODE 1, small alphabets: the contraction axis is split over blockIdx.z (a 128 x 128 output has ONE tile: the whole
    // product would run on one compute unit, 1.7 ms at N = 128 B T = 25 600); slice z leaves its raw sums in partial[z],
    // gemm_combine_kernel adds the slices in order and applies the E factor
    // BK = 32: one stage of global -> register -> LDS staging (and its two workgroup barriers) per 128 MFMAs of a
    // wavefront; at BK = 16 the barriers and the LDS round trip took 29 % of the kernel (112 of 157 TFLOP/s)
    constexpr int BK = ASG_X_GEMM_BK, TS = 128, LD = TS + 4, NST = BK * TS / 4 / 256;      // NST float4 per thread and operand
    __shared__ __attribute__((aligned(16))) R As[BK][LD];
    __shared__ __attribute__((aligned(16))) R Bs[BK][LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int m0 = blockIdx.x * TS, n0 = blockIdx.y * TS;
    const int Mdim = N, Ndim = MODE == 0 ? K : N;
    const int kbeg = (MODE == 1 && partial) ? (int) blockIdx.z * kslice : 0;
    const int Kdim = (MODE == 1 && partial) ? min(K, kbeg + kslice) : (MODE == 0 ? npad : K);
    const V4f zero4 = {0, 0, 0, 0};
    V4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = zero4;
    // staging: two float4 of A and two of B per thread and tile
    //   MODE 0: rows of ehat / Pm (k contiguous): element e -> row e >> 2, k quad e & 3      (transposed into As[k][m])
    //   MODE 1: rows of Gm / Pm (m / n contiguous): element e -> k row e >> 5, column quad e & 31
    auto fetchA = [&](int k0, int e) -> V4f {
        if (MODE == 0) {
            const int mm = m0 + e / (BK / 4), kk = k0 + 4 * (e % (BK / 4));
            return (mm < Mdim && kk < Kdim) ? *reinterpret_cast<const V4f *>(ehat + (int64_t) mm * npad + kk) : zero4;
        } else {
            const int kk = k0 + (e >> 5), mm = m0 + 4 * (e & 31);
            V4f v = (kk < Kdim && mm < npad) ? *reinterpret_cast<const V4f *>(Gm + (int64_t) kk * npad + mm) : zero4;
            const float ninf = -__builtin_inff();                      // markers of the row-sum pass read as 0
            v.x = v.x == ninf ? 0.f : v.x; v.y = v.y == ninf ? 0.f : v.y; v.z = v.z == ninf ? 0.f : v.z; v.w = v.w == ninf ? 0.f : v.w;
            return v;
        }
    };
    auto fetchB = [&](int k0, int e) -> V4f {
        if (MODE == 0) {
            const int nn = n0 + e / (BK / 4), kk = k0 + 4 * (e % (BK / 4));
            return (nn < Ndim && kk < Kdim) ? *reinterpret_cast<const V4f *>(Pm + (int64_t) nn * npad + kk) : zero4;
        } else {
            const int kk = k0 + (e >> 5), nn = n0 + 4 * (e & 31);
            return (kk < Kdim && nn < npad) ? *reinterpret_cast<const V4f *>(Pm + (int64_t) kk * npad + nn) : zero4;
        }
    };
    auto put = [&](R (*dst)[LD], int e, const V4f &v) {
        if (MODE == 0) {
            const int mm = e / (BK / 4), kq = 4 * (e % (BK / 4));
            dst[kq + 0][mm] = v.x; dst[kq + 1][mm] = v.y; dst[kq + 2][mm] = v.z; dst[kq + 3][mm] = v.w;
        } else {
            *reinterpret_cast<V4f *>(&dst[e >> 5][4 * (e & 31)]) = v;
        }
    };
    V4f sa[NST], sb[NST];
#pragma unroll
    for (int r = 0; r < NST; ++r) { sa[r] = fetchA(kbeg, (int) threadIdx.x + 256 * r); sb[r] = fetchB(kbeg, (int) threadIdx.x + 256 * r); }
    for (int k0 = kbeg; k0 < Kdim; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NST; ++r) { put(As, (int) threadIdx.x + 256 * r, sa[r]); put(Bs, (int) threadIdx.x + 256 * r, sb[r]); }
        __syncthreads();
        if (k0 + BK < Kdim) {
#pragma unroll
            for (int r = 0; r < NST; ++r) { sa[r] = fetchA(k0 + BK, (int) threadIdx.x + 256 * r); sb[r] = fetchB(k0 + BK, (int) threadIdx.x + 256 * r); }
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[ks + (lane >> 4)][wm + 16 * a + (lane & 15)];
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = Bs[ks + (lane >> 4)][wn + 16 * c + (lane & 15)];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[c], acc[a][c], 0, 0, 0);
        }
    }
    // element (m = 16 a + 4 (l >> 4) + q, n = 16 c + (l & 15)) of the wavefront's 64 x 64 sits in acc[a][c][q]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int gm0 = m0 + wm + 16 * a + 4 * (lane >> 4), gn = n0 + wn + 16 * c + (lane & 15);
            if (gn >= Ndim) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gm_ = gm0 + q;
                if (gm_ >= Mdim) continue;
                if (MODE == 0) {
                    const R g = Gm[(int64_t) gn * npad + gm_];
                    const R sden = acc[a][c][q];
                    const bool ok = fabs(Num<R>::log2(sden)) < Num<R>::lg_limit();
                    const bool mark = !ok && g != R(0);
                    Gm[(int64_t) gn * npad + gm_] = ok ? g / sden : (mark ? Num<R>::ninf() : R(0));
                    if (mark) *anybad = 1;
                } else if (partial) {
                    partial[((int64_t) blockIdx.z * N + gm_) * N + gn] = acc[a][c][q];
                } else {
                    out[(int64_t) gm_ * N + gn] = acc[a][c][q] * ehat[(int64_t) gm_ * npad + gn];
                }
            }
        }
}

// ---- the same contraction on the bfloat16 matrix pipe, fp32-equivalent (round 5; large alphabets: one slice of the frame axis) ----
// G = U^T P over K ~ 48 000 valid frame rows at cfg 5 is 9.6 TFLOP: at the fp32 matrix rate (v_mfma_f32_16x16x4_f32 = the vector
// rate, 157 TFLOP/s) 61 ms at best, 85 ms as measured.  Every float is EXACTLY the sum of three bfloat16 (8 + 8 + 8 significant
// bits, round to nearest at each step, remainders exact), and the six partial products of weight >= 2^-16
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi           (dropped: mid*lo + lo*mid + lo*lo <= 2^-24 relative, zero mean)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 cost 6 x 32 cycles per 32 x 32 x 16 block where the fp32 instruction takes
// 16 x 32: 2.7x less matrix-pipe time at the accuracy of an fp32 product chain.  gemm3_pack_kernel splits the two operands ONCE
// (each element is used by ~80 output tiles) into planes laid out [K/8][npadT][8]: a lane's eight consecutive k of one label are
// 16 contiguous bytes -- the instruction's operand as it is, in memory, in LDS and in registers.
constexpr int kG3TM = 256, kG3TN = 256;          // output tile of a workgroup (8 wavefronts, 64 x 128 each)
__host__ __device__ inline int g3_npadT(int N) { return (N + kG3TM - 1) / kG3TM * kG3TM; }
__host__ __device__ inline size_t g3_plane_elems(int K, int N) { return (size_t) ((K + 31) / 32 * 32) * g3_npadT(N); }

// grid = (npadT / 256, ceil(Kmax / 8)), block = 256: thread = label m, rows 8 kg .. + 7 of X [K][npad] (-inf markers and rows >= K: 0)
__global__ void __launch_bounds__(256) gemm3_pack_kernel(const float *X, int npad, int npadT, const int *kdev, int K, unsigned short *planes,
                                                         size_t plane_elems) {
    if (kdev) K = __builtin_amdgcn_readfirstlane(*kdev);
    const int kg = blockIdx.y;
    if (8 * kg >= (K + 31) / 32 * 32) return;
    const int m = blockIdx.x * 256 + threadIdx.x;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = 8 * kg + q;
        float x = (k < K && m < npad) ? X[(int64_t) k * npad + m] : 0.f;
        v[q] = (x == -__builtin_inff()) ? 0.f : x;
    }
    unsigned h[4], mi[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split3x2(v[2 * q], v[2 * q + 1], h[q], mi[q], lo[q]);
    U4v *dst = reinterpret_cast<U4v *>(planes) + ((size_t) kg * npadT + m);
    const size_t pu = plane_elems / 8;          // 16-byte units per plane
    dst[0] = U4v{h[0], h[1], h[2], h[3]};
    dst[pu] = U4v{mi[0], mi[1], mi[2], mi[3]};
    dst[2 * pu] = U4v{lo[0], lo[1], lo[2], lo[3]};
}

typedef float V16f __attribute__((ext_vector_type(16)));
// grid = 8 x 32 x ceil(blocks / 8) workgroups (1-D), block = 512, dynamic LDS = 2 stages x 72 KB.
// Workgroup id -> tile: id & 7 is the XCD the dispatcher puts it on; an XCD walks blocks of 4 x 8 tiles (1024 labels x 1024 labels of
// output: its 32 resident workgroups share 4 row panels and 8 column panels through that XCD's L2).
// grid = 8 x 32 x ceil(blocks / 8) workgroups (1-D), block = 512 (8 wavefronts, 64 x 128 of the 256 x 256 tile each), dynamic LDS =
// 3 stages x 48 KB (16 k per stage: one v_mfma_f32_32x32x16_bf16 step).
// Workgroup id -> tile: id & 7 is the XCD the dispatcher puts it on; an XCD walks blocks of 4 x 8 tiles (its 32 resident workgroups share
// 4 row panels and 8 column panels through that XCD's L2).
// What bounded the first form (256 x 128 tiles, 64 x 64 per wavefront, 32 k per stage: 49 ms at cfg 5 where the products alone take 34 and
// the staging alone 25-32) was LDS traffic and transfer issue per matrix instruction: 24 fragment reads and 18 transfers per 48 products of
// a wavefront.  A 64 x 128 wavefront tile reads 18 fragments per 48 products, and a 256 x 256 workgroup tile needs 12 transfers per loader
// wavefront for them; three stages give a transfer two steps to land.
constexpr int kG3Stage = 3 * 2 * (kG3TM + kG3TN);          // 16-byte units per stage: 3 planes x 2 k groups x (256 + 256) labels
constexpr size_t kG3LdsBytes = (size_t) 3 * kG3Stage * 16;
__global__ void __launch_bounds__(512) bwd_gemm_bf3_kernel(const unsigned short *Apl, const unsigned short *Bpl, size_t plane_elems,
                                                           const float *ehat, float *out, int N, int npad, int npadT, const int *kdev, int K,
                                                           int Mt, int Nt, int block0, int ks, float *partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char g3_lds[];
    U4v *lds = reinterpret_cast<U4v *>(g3_lds);
    constexpr int TM = kG3TM, TN = kG3TN, AU = 3 * 2 * TM, SU = kG3Stage;
    constexpr int ND = SU / 64 / 4;                                  // transfers per loader wavefront and stage: 12
    if (kdev) K = __builtin_amdgcn_readfirstlane(*kdev);
    // (the LAST, partial round of blocks is a launch of its own with the frame axis cut into ks slices, so that its few tiles occupy
    // the whole device too: slice s of tile (g, r) leaves raw sums in partial[((g - block0) * 32 + r) * ks + s], gemm3_tail_kernel adds them)
    const int slice = (int) blockIdx.x % ks, id = (int) blockIdx.x / ks, xcd = id & 7, j = id >> 3;
    const int mblocks = (Mt + 3) / 4, nblocks = (Nt + 7) / 8;
    // (the sliced launch is compact: tile id of the tail = id, no workgroup without work)
    const int g = ks > 1 ? block0 + (id >> 5) : block0 + (j >> 5) * 8 + xcd, r = ks > 1 ? (id & 31) : (j & 31);
    if (g >= mblocks * nblocks) return;
    const int tm = (g % mblocks) * 4 + (r & 3), tn = (g / mblocks) * 8 + (r >> 2);
    if (tm >= Mt || tn >= Nt) return;
    const int m0 = tm * TM, n0 = tn * TN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int wm = (wave & 3) * 64, wn = (wave >> 2) * 128;
    const size_t pu = plane_elems / 8;
    const U4v *Au = reinterpret_cast<const U4v *>(Apl), *Bu = reinterpret_cast<const U4v *>(Bpl);
    // Staging is LDS-DMA (64 lanes x 16 bytes land lane-linear at a wave-uniform LDS address -- the planes' [k group][label][8] order IS
    // the stage's order, so no register or ds_write is involved), issued by wavefronts 0-3 only: each SIMD holds wavefronts w and w + 4;
    // while w issues its transfers, w + 4 has the matrix pipe to itself, then both interleave (four more wavefronts that do nothing but
    // transfers: measured slower, 60 against 55 ms).  In-kernel probe (-DASG_X_G3_PROBE, cycles per 16-k step at cfg 5): fragment reads
    // 330-580, transfers 660 (2055 as global_load_lds), the SIMD's 96 products 3043 = 31.7 apiece back to back, drain + barrier ~270: the
    // matrix pipe is busy 78 % of the step; the rest is the LDS serving 144 KB of fragment reads to eight wavefronts at the step's start.
    const bool loader = wave < 4;
    // transfer d of a loader wavefront (12 per stage): operand d / 6, plane (d % 6) / 2, k group d % 2, labels 64 (wave & 3) + lane of the
    // tile -- as RAW BUFFER loads (buffer_load_dwordx4 .. offen lds): one descriptor per (operand, plane) in scalar registers, the lane's byte
    // offset in ONE vector register per operand for the whole kernel, the step's row offset a scalar.  (global_load_lds_dwordx4 needs a
    // 64-bit address per lane and transfer: two vector adds, a readfirstlane and four scalar moves around every one of them, 171 cycles of
    // issue per transfer by the in-kernel probe -- and the twelve transfers in front of a loader's products are the step's critical path.)
    const unsigned rowbytes = (unsigned) npadT * 16u;          // one k group of one plane
    const unsigned planebytes = (unsigned) (pu * 16);          // (the launcher takes this route only while a plane stays below 4 GB)
    __amdgpu_buffer_rsrc_t rsA[3], rsB[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        rsA[pl] = __builtin_amdgcn_make_buffer_rsrc((void *) (Au + (size_t) pl * pu), 0, planebytes, 0x00020000);
        rsB[pl] = __builtin_amdgcn_make_buffer_rsrc((void *) (Bu + (size_t) pl * pu), 0, planebytes, 0x00020000);
    }
    const unsigned vA = (unsigned) (m0 + 64 * (wave & 3) + lane) * 16u, vB = (unsigned) (n0 + 64 * (wave & 3) + lane) * 16u;
    const int nall = (K + 31) / 32 * 2;                 // steps of 16 k (the planes are zero-padded to whole 32-row blocks)
    const int per = (nall + ks - 1) / ks, first = min(slice * per, nall);
    const int nst = min(first + per, nall) - first;     // this workgroup's steps: first .. first + nst
    V16f acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][c][q] = 0.f;
    auto dma = [&](int st) {
        // (inline asm: hipcc counts an LDS-DMA builtin against EVERY later LDS read -- s_waitcnt vmcnt(0) in front of the fragment reads
        // of the stage being multiplied, which is not the stage being filled; the drains are the explicit ones below.  M0 = LDS address.)
        const unsigned base = (unsigned) (uintptr_t) (__attribute__((address_space(3))) void *) (lds + (st % 3) * SU) + 1024u * (unsigned) (wave & 3);
        const unsigned srow = (unsigned) (first + st) * 2u * rowbytes;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            // (readfirstlane: hipcc keeps the row offset in a vector register otherwise; s_nop 4: a scalar register written by the vector
            // ALU needs five wait states before a buffer instruction reads it as soffset)
            const unsigned l = __builtin_amdgcn_readfirstlane(base + 4096u * d), so = __builtin_amdgcn_readfirstlane(srow + (d & 1 ? rowbytes : 0u));
            if (d < ND / 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(vA), "s"(rsA[(d % 6) / 2]), "s"(l), "s"(so) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(vB), "s"(rsB[(d % 6) / 2]), "s"(l), "s"(so) : "memory", "m0");
        }
    };
    if (loader) {
        if (nst > 0) dma(0);
        if (nst > 1) dma(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#ifdef ASG_X_G3_PROBE
    long long tq[5] = {0, 0, 0, 0, 0};          // fragment reads | transfer issue | products | drain | barrier
#define G3_T(i, expr) { const long long _a = __builtin_readcyclecounter(); expr; tq[i] += (long long) __builtin_readcyclecounter() - _a; }
#else
#define G3_T(i, expr) { expr; }
#endif
    for (int st = 0; st < nst; ++st) {
        const U4v *cur = lds + (st % 3) * SU;
        const int kg = lane >> 5, ln = lane & 31;
        BF8 af[2][3], bf[4][3];
#ifdef ASG_X_G3_PROBE
        const long long t_f0 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[a][pl] = __builtin_bit_cast(BF8, cur[pl * 2 * TM + kg * TM + wm + 32 * a + ln]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[c][pl] = __builtin_bit_cast(BF8, cur[AU + pl * 2 * TN + kg * TN + wn + 32 * c + ln]);
#ifdef ASG_X_G3_PROBE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tq[0] += (long long) __builtin_readcyclecounter() - t_f0;
#endif
        // (pinned: an asm statement orders memory operations only -- left alone, hipcc lifts the drain + barrier above half of the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        // into the stage every wavefront finished reading before the last barrier
#if defined(ASG_X_G3_ABL) && ASG_X_G3_ABL == 1          // (developer timing: no transfers inside the loop, wrong results)
        const bool issue = false;
#else
        const bool issue = loader && st + 2 < nst;
#endif
        G3_T(1, if (issue) dma(st + 2);)
        __builtin_amdgcn_sched_barrier(0);
#ifdef ASG_X_G3_PROBE
        const long long t_m0 = __builtin_readcyclecounter();
#endif
#if defined(ASG_X_G3_ABL) && ASG_X_G3_ABL == 2          // (developer timing: transfers and fragment reads only, wrong results)
        if (st == 0)
#elif defined(ASG_X_G3_ABL) && ASG_X_G3_ABL == 3        // (developer timing: a third of the products)
        if (st % 3 == 0)
#endif
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // (smallest terms first)
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[c][0], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[c][2], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[c][1], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[c][0], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[c][1], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[c][0], acc[a][c], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
#ifdef ASG_X_G3_PROBE
        asm volatile("s_nop 0" : "+v"(acc[1][3]));          // (the last product has issued)
        tq[2] += (long long) __builtin_readcyclecounter() - t_m0;
#endif
        // the NEXT step's stage has landed (the transfers issued in this step may still travel)
        G3_T(3, if (issue) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");)
        G3_T(4, __syncthreads();)
    }
#ifdef ASG_X_G3_PROBE
    if (blockIdx.x == 17 && lane == 0 && ks == 1)
        printf("[g3] wave %d: %d steps; cycles per step: fragments %lld, transfer issue %lld, products %lld, drain %lld, barrier %lld\n", wave, nst,
               tq[0] / nst, tq[1] / nst, tq[2] / nst, tq[3] / nst, tq[4] / nst);
#endif
    // element (m = 32 a + 8 (q >> 2) + 4 (l >> 5) + (q & 3), n = 32 c + (l & 31)) of the wavefront's 64 x 128 sits in acc[a][c][q]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ln = wn + 32 * c + (lane & 31), gn = n0 + ln;
            if (ks > 1) {
                float *pt = partial + ((size_t) ((g - block0) * 32 + r) * ks + slice) * (TM * TN);
#pragma unroll
                for (int q = 0; q < 16; ++q) pt[(size_t) (wm + 32 * a + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3)) * TN + ln] = acc[a][c][q];
                continue;
            }
            if (gn >= N) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int gm_ = m0 + wm + 32 * a + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3);
                if (gm_ < N) out[(int64_t) gm_ * N + gn] = acc[a][c][q] * ehat[(int64_t) gm_ * npad + gn];
            }
        }
}

// the sliced tail tiles: out = ehat o (slice 0 + slice 1 + ...), ascending.  grid = (tail blocks * 32, TM * TN / 1024), block = 256 (float4 each)
__global__ void __launch_bounds__(256) gemm3_tail_kernel(const float *partial, const float *ehat, float *out, int N, int npad, int Mt, int Nt,
                                                         int block0, int ks) {
    const int t = blockIdx.x, g = block0 + (t >> 5), r = t & 31;
    const int mblocks = (Mt + 3) / 4;
    const int tm = (g % mblocks) * 4 + (r & 3), tn = (g / mblocks) * 8 + (r >> 2);
    if (tm >= Mt || tn >= Nt) return;
    const int e = ((int) blockIdx.y * 256 + (int) threadIdx.x) * 4, row = e / kG3TN, col = e % kG3TN;
    const int gm_ = tm * kG3TM + row, gn = tn * kG3TN + col;
    if (gm_ >= N) return;
    V4f sum = {0, 0, 0, 0};
    for (int k = 0; k < ks; ++k) sum += *reinterpret_cast<const V4f *>(partial + ((size_t) t * ks + k) * (kG3TM * kG3TN) + e);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (gn + q < N) out[(int64_t) gm_ * N + gn + q] = sum[q] * ehat[(int64_t) gm_ * npad + gn + q];
}

// out[m][n] = ehat[m][n] * sum over the slices of partial[z][m][n], slices in ascending order.  grid = ceil(N^2 / 256).
template <typename R>
__global__ void __launch_bounds__(256) gemm_combine_kernel(const R *partial, int nslices, const R *ehat, int N, int npad, R *out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N * N) return;
    const int m = k / N, n = k - m * N;
    R a[4] = {0, 0, 0, 0};
    int z = 0;
    for (; z + 4 <= nslices; z += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += partial[(int64_t) (z + q) * N * N + k];
    }
    for (; z < nslices; ++z) a[0] += partial[(int64_t) z * N * N + k];
    out[k] = ((a[0] + a[1]) + (a[2] + a[3])) * ehat[(int64_t) m * npad + n];
}

// exact fix-up of marked rows (rare; exits at once unless the row-sum pass raised `anybad`).
// grid = (T, B), block = 256: recomputes the posterior of each marked (b,t,i) and adds
// gi * softmax_j(Tr2[i][j] + ah[t-1][j]) into grad_transition.  Different (b,t) can hit the same (i,j), so this
// path uses a float atomicAdd: it only runs for degenerate inputs (transition spans > 69 nats) and is the one
// place whose summation ORDER is not fixed.
template <typename R>
__global__ void __launch_bounds__(256) bwd_fix_kernel(Problem P, State W, BwdArgs A, R *Gm, R *out, int npad, const int *anybad,
                                                      const int *rowoff) {
    __shared__ R red[4];
    if (!*anybad) return;
    const int t = blockIdx.x, b = blockIdx.y, N = P.N, T = P.T;
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    if (t < 1 || t >= len) return;
    R *gm = Gm + (rowoff ? (int64_t) rowoff[b] + t - 1 : (int64_t) b * T + t) * npad;
    const R *ah = (const R *) W.ah + ((int64_t) b * T + t) * N;
    const R *bh = (const R *) W.bh + ((int64_t) b * T + t) * N;
    const R *ahp = ah - N;
    const R *tr = (const R *) P.transition;
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const R gf = (R) ((double) ((const R *) A.grad_full)[(int64_t) b * A.gstride] * A.gscale);
    R m = Num<R>::ninf();
    for (int i = threadIdx.x; i < N; i += 256) m = fmax(m, ah[i] + bh[i]);
    m = fmax(block_reduce_max<R>(m, red), LZ);
    R z = 0;
    for (int i = threadIdx.x; i < N; i += 256) z += Num<R>::exp2(ah[i] + bh[i] - m);
    z = block_reduce_sum<R>(z, red);
    for (int i = 0; i < N; ++i) {
        if (!(gm[i] == Num<R>::ninf())) continue;             // uniform: every thread reads the same element
        R g = (z > 0) ? gf * Num<R>::exp2(ah[i] + bh[i] - m) / z : R(0);
        R mx = Num<R>::ninf();
        for (int j = threadIdx.x; j < N; j += 256) {
            R v = tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1] * L2E + ahp[j];
            mx = (v == v) ? fmax(mx, v) : mx;
        }
        mx = block_reduce_max<R>(mx, red);
        if (mx == Num<R>::ninf()) continue;
        R sm = 0;
        for (int j = threadIdx.x; j < N; j += 256) {
            R v = tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1] * L2E + ahp[j];
            sm += (v == v) ? Num<R>::exp2(v - mx) : R(0);
        }
        sm = block_reduce_sum<R>(sm, red);
        for (int j = threadIdx.x; j < N; j += 256) {
            R v = tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1] * L2E + ahp[j];
            R x = (v == v) ? g * Num<R>::exp2(v - mx) / sm : R(0);
            if (x != R(0)) atomicAdd(&out[(int64_t) i * N + j], x);
        }
    }
}


}  // namespace

// slices of the frame axis for the outer-product contraction: enough workgroups to fill the device when the output
// has only a few 128 x 128 tiles (N <= 1024), each slice at least 256 rows long
static int gemm_slices(int N, int K) {
    const int tiles = ((N + 127) / 128) * ((N + 127) / 128);
    if (tiles >= 512) return 1;          // (64 tiles = N of 1024 ran on a quarter of the chip: 1.9 ms where 8 slices take 0.5)
    int n = (512 + tiles - 1) / tiles;
    if (n > K / 256) n = K / 256;
    return n < 1 ? 1 : n;
}

// the contraction's operands as three bfloat16 planes each (large alphabets, fp32, one slice of the frame axis): bytes per operand
#ifndef ASG_X_GEMM_BF3
#define ASG_X_GEMM_BF3 1
#endif
constexpr int kG3TailSlices = 4, kG3WholeSlices = 8;      // slices of a partial last round / of a grid that is a single partial round
constexpr size_t kG3TailBytes = (size_t) 208 << 20;          // 128 tail tiles x 4 slices x 256 KB; or a whole single round (8 blocks x 32 tiles) x 3 slices
#ifndef ASG_X_G3_MIN_N
#define ASG_X_G3_MIN_N 1024
#endif
constexpr int kG3MinN = ASG_X_G3_MIN_N;
static size_t gemm3_plane_bytes(int elem, int T, int B, int N) {
    // (beyond 1024 labels; round 5 first took it only where the fp32 contraction ran unsliced, N >= ~2900 -- below that the grid is a
    // single partial round of 256 x 256 tiles, which the sliced launch now fills: N = 1500 36 tiles x 7 slices)
    if (!(ASG_X_GEMM_BF3 && elem == 4 && StepUsesMfma<float>::v && N > kG3MinN)) return 0;
    if ((double) g3_plane_elems(B * T, N) * 2.0 >= 4294967296.0) return 0;          // (a plane is addressed through one 32-bit buffer resource)
    return au(3 * g3_plane_elems(B * T, N) * sizeof(unsigned short));
}

size_t bwd_scratch_bytes_generic(int elem, int T, int B, int N, int S) {
    const size_t npad = (size_t) (N + 3) / 4 * 4;
    int ch, nch;
    generic_chunks(T, B, &ch, &nch);
    size_t tiles = N <= 64 ? au((size_t) B * nch * N * N * elem) : 0;             // bwd_aligned_long_kernel
    if (N > 64 && N <= 2048) tiles = au((size_t) N * N * 8);                      // aligned_tr_scatter_fx_kernel
    if (S > 1024 && N <= 2048 && tiles < au((size_t) N * N * 8)) tiles = au((size_t) N * N * 8);      // (very long targets: the same accumulator for any N <= 2048)
    if (N > 64) tiles += au((size_t) gemm_slices(N, B * T) * N * N * elem);       // split contraction: partial sums
    tiles += 2 * gemm3_plane_bytes(elem, T, B, N);                                  // bfloat16 planes of both operands (bwd_gemm_bf3_kernel)
    if (gemm3_plane_bytes(elem, T, B, N)) tiles += kG3TailBytes;                    // ... and the sliced tiles of its last, partial round
    return 2 * au((size_t) B * T * npad * elem) + au((size_t) B * nch * 2 * S * elem) + 512 + au(((size_t) B + 1) * 4) + tiles;
}

GenericBwdLayout generic_bwd_layout(size_t e, const Problem &P, const BwdArgs &A) {
    GenericBwdLayout Y{};
    const int npad = (P.N + 3) / 4 * 4;
    char *sc = (char *) A.scratch;
    Y.npad = npad;
    Y.Pm = sc; sc += au((size_t) P.B * P.T * npad * e);
    Y.Gm = sc; sc += au((size_t) P.B * P.T * npad * e);
    Y.gHD = sc; sc += au((size_t) P.B * A.nchunks * 2 * P.S * e);
    Y.anybad = (int *) sc; sc += 512;
    Y.rowoff = (int *) sc; sc += au(((size_t) P.B + 1) * 4);
    Y.atiles = sc;
    {
        size_t tb = P.N <= 64 ? au((size_t) P.B * A.nchunks * P.N * P.N * e) : (P.N <= 2048 ? au((size_t) P.N * P.N * 8) : 0);
        if (P.S > 1024 && P.N <= 2048 && tb < au((size_t) P.N * P.N * 8)) tb = au((size_t) P.N * P.N * 8);      // (as bwd_scratch_bytes_generic)
        sc += tb;
    }
    Y.gpart = sc;
    if (P.N > 64) sc += au((size_t) gemm_slices(P.N, P.B * P.T) * P.N * P.N * e);
    Y.planes3 = (unsigned short *) sc;      // (only when gemm3_plane_bytes says so)
    return Y;
}

template <typename R>
hipError_t launch_bwd_full_generic(const Problem &P, const State &W, const BwdArgs &A, const GenericBwdLayout &Y, bool do_ali, bool *fx_cleared_out,
                                   hipStream_t stream) {
    const size_t e = sizeof(R);
    const int npad = Y.npad;
    R *Pm = (R *) Y.Pm, *Gm = (R *) Y.Gm, *atiles = (R *) Y.atiles, *gpart = (R *) Y.gpart, *gtr = (R *) A.grad_transition;
    int *anybad = Y.anybad, *rowoff = Y.rowoff;
    unsigned short *planes3 = Y.planes3;
    bool fx_cleared = false;
    const bool do_full = true;
    if (do_full) {
        if (P.N <= 64) return hipErrorInvalidValue;      // the small kernel owns this case
        const int K = P.B * P.T;
        // one clear for the flag word and, when the aligned part follows with its fixed-point scatter buffer (64 < N <= 2048), for
        // that buffer too: only the row-offset table lies between them, and it is written later on this stream
        fx_cleared = do_ali && P.N > 64 && P.N <= 2048;
        (void) zero_async(anybad, fx_cleared ? (size_t) ((char *) atiles - (char *) anybad) + (size_t) P.N * P.N * 8 : sizeof(int), stream);
        const R *emax = (const R *) W.work;
        const R *mulog = (const R *) ((const char *) W.work + work_mulog_offset(e, P.T, P.B, npad));
        // (medium alphabets, fwd_mid_kernel, log the same per-frame normaliser as the streamed step since round 3: one branch)
        {
        if constexpr (StepUsesMfma<R>::v) {
            if (!W.work) return hipErrorInvalidValue;
            hipLaunchKernelGGL(rowoff_kernel, dim3(1), dim3(64), 0, stream, P, rowoff);
            hipLaunchKernelGGL((bwd_post_kernel<R, true>), dim3(P.T, P.B), dim3(256), 0, stream, P, W, A, Pm, Gm, npad, emax, mulog, anybad,
                               (const int *) rowoff);
        } else {
            // fp64: the row sums from the stored state as well (every double-precision forward logs its normaliser too; round 4) --
            // rows in place (no compaction: the VALU contraction below takes K = B T)
            if (!W.work) return hipErrorInvalidValue;
            hipLaunchKernelGGL((bwd_post_kernel<R, true>), dim3(P.T, P.B), dim3(256), 0, stream, P, W, A, Pm, Gm, npad, emax, mulog, anybad,
                               (const int *) nullptr);
        }
        if constexpr (StepUsesMfma<R>::v) {
            // (no row-sum contraction: bwd_post_kernel<.., true> derived the row sums from the stored state)
            const int tiles1 = ((P.N + 127) / 128) * ((P.N + 127) / 128);
            const int nsl = gemm_slices(P.N, K);
            const size_t pbytes3 = gemm3_plane_bytes((int) e, P.T, P.B, P.N);
            if (nsl > 1 && !pbytes3) {
                const int kslice = ((K + nsl - 1) / nsl + ASG_X_GEMM_BK - 1) / ASG_X_GEMM_BK * ASG_X_GEMM_BK;
                hipLaunchKernelGGL((bwd_gemm_mfma<1>), dim3((P.N + 127) / 128, (P.N + 127) / 128, nsl), dim3(256), 0, stream,
                                   (const float *) W.ehat, (const float *) Pm, (float *) Gm, (float *) gtr, P.N, npad, K, anybad,
                                   (const int *) (rowoff + P.B), kslice, (float *) gpart);
                hipLaunchKernelGGL((gemm_combine_kernel<float>), dim3((P.N * P.N + 255) / 256), dim3(256), 0, stream, (const float *) gpart,
                                   nsl, (const float *) W.ehat, P.N, npad, (float *) gtr);
            } else if (const size_t pbytes = pbytes3) {
                // large alphabets: both operands split into bfloat16 planes once, the product on v_mfma_f32_32x32x16_bf16
                const int npadT = g3_npadT(P.N);
                const size_t pe = g3_plane_elems(K, P.N);
                unsigned short *apl = planes3, *bpl = (unsigned short *) ((char *) planes3 + pbytes);
                const dim3 pgrid(npadT / 256, (K + 31) / 32 * 4);          // whole 32-row blocks: the product reads four 8-row groups per block
                hipLaunchKernelGGL(gemm3_pack_kernel, pgrid, dim3(256), 0, stream, (const float *) Gm, npad, npadT, (const int *) (rowoff + P.B), K, apl, pe);
                hipLaunchKernelGGL(gemm3_pack_kernel, pgrid, dim3(256), 0, stream, (const float *) Pm, npad, npadT, (const int *) (rowoff + P.B), K, bpl, pe);
                const int Mt = npadT / kG3TM, Nt = (P.N + kG3TN - 1) / kG3TN;
                const int blocks = ((Mt + 3) / 4) * ((Nt + 7) / 8);
                const size_t lds = kG3LdsBytes;
                (void) hipFuncSetAttribute((const void *) bwd_gemm_bf3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
                // whole rounds of 8 blocks (one per XCD, 32 tiles each) in one launch; a partial last round that would leave most of
                // the device idle for a whole tile time (cfg 5: 50 blocks = 6 rounds + 2 blocks) as a second launch, the frame axis sliced
                int tail = blocks % 8, tks = 1;
                if (blocks <= 8) {
                    // a single, partial round (N = 3000: 144 tiles on 256 compute units): the whole product goes the sliced way when that
                    // takes fewer tile times -- rounds of the sliced grid / slices (N = 3000: 3 slices, 432 workgroups = 2 rounds of a third)
                    tail = blocks;
                    const int real = Mt * Nt;
                    double best = (double) ((real + 255) / 256);
                    for (int t = 2; t <= kG3WholeSlices; ++t) {
                        const double c = (double) ((real * t + 255) / 256) / t;
                        if (c < best - 0.05 && (size_t) tail * 32 * t * kG3TM * kG3TN * sizeof(float) <= kG3TailBytes) { best = c; tks = t; }
                    }
                } else if (tail * 32 * 2 <= 256 && tail > 0) {
                    tks = 256 / (tail * 32) > kG3TailSlices ? kG3TailSlices : 256 / (tail * 32);
                }
                if (tks < 2 || (size_t) tail * 32 * tks * kG3TM * kG3TN * sizeof(float) > kG3TailBytes) { tail = 0; tks = 1; }
                const int mainb = blocks - tail;
                float *tpart = (float *) ((char *) planes3 + 2 * pbytes);
                if (mainb > 0)
                    hipLaunchKernelGGL(bwd_gemm_bf3_kernel, dim3(8 * 32 * ((mainb + 7) / 8)), dim3(512), lds, stream, (const unsigned short *) apl,
                                       (const unsigned short *) bpl, pe, (const float *) W.ehat, (float *) gtr, P.N, npad, npadT,
                                       (const int *) (rowoff + P.B), K, Mt, Nt, 0, 1, (float *) nullptr);
                if (tail) {
                    hipLaunchKernelGGL(bwd_gemm_bf3_kernel, dim3(tail * 32 * tks), dim3(512), lds, stream, (const unsigned short *) apl,
                                       (const unsigned short *) bpl, pe, (const float *) W.ehat, (float *) gtr, P.N, npad, npadT,
                                       (const int *) (rowoff + P.B), K, Mt, Nt, mainb, tks, tpart);
                    hipLaunchKernelGGL(gemm3_tail_kernel, dim3(tail * 32, kG3TM * kG3TN / 1024), dim3(256), 0, stream, (const float *) tpart,
                                       (const float *) W.ehat, (float *) gtr, P.N, npad, Mt, Nt, mainb, tks);
                }
            } else {
                hipLaunchKernelGGL((bwd_gemm_mfma<1>), dim3((P.N + 127) / 128, (P.N + 127) / 128), dim3(256), 0, stream,
                                   (const float *) W.ehat, (const float *) Pm, (float *) Gm, (float *) gtr, P.N, npad, K, anybad,
                                   (const int *) (rowoff + P.B), 0, (float *) nullptr);
            }
            (void) tiles1;
        } else {
            // (no row-sum contraction either: bwd_gemm_kernel<R, 0> was 0.5 / 1.9 / 7 ms at N = 512 / 1024 / 2048, T = 400, B = 64)
            const int nsl = gemm_slices(P.N, K);
            if (nsl > 1) {
                const int kslice = ((K + nsl - 1) / nsl + 15) / 16 * 16;
                hipLaunchKernelGGL((bwd_gemm_kernel<R, 1>), dim3((P.N + 63) / 64, (P.N + 63) / 64, nsl), dim3(256), 0, stream,
                                   (const R *) W.ehat, Pm, Gm, gtr, P.N, npad, K, anybad, kslice, gpart);
                hipLaunchKernelGGL((gemm_combine_kernel<R>), dim3((P.N * P.N + 255) / 256), dim3(256), 0, stream, (const R *) gpart,
                                   nsl, (const R *) W.ehat, P.N, npad, gtr);
            } else {
                hipLaunchKernelGGL((bwd_gemm_kernel<R, 1>), dim3((P.N + 63) / 64, (P.N + 63) / 64), dim3(256), 0, stream,
                                   (const R *) W.ehat, Pm, Gm, gtr, P.N, npad, K, anybad);
            }
        }
        hipLaunchKernelGGL((bwd_fix_kernel<R>), dim3(P.T, P.B), dim3(256), 0, stream, P, W, A, Gm, gtr, npad, anybad,
                           StepUsesMfma<R>::v ? (const int *) rowoff : (const int *) nullptr);
        }
    }    if (fx_cleared_out) *fx_cleared_out = fx_cleared;
    return hipGetLastError();
}

template hipError_t launch_bwd_full_generic<float>(const Problem &, const State &, const BwdArgs &, const GenericBwdLayout &, bool, bool *, hipStream_t);
template hipError_t launch_bwd_full_generic<double>(const Problem &, const State &, const BwdArgs &, const GenericBwdLayout &, bool, bool *, hipStream_t);

}  // namespace asg
