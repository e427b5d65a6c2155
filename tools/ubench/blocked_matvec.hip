// Developer micro-benchmark (gfx950): the N x N mat-vec of one recursion step WITHOUT the LDS broadcast, as a 2-D blocked
// product over an 8 x 8 grid of lanes.  lane = 8 r + c.  BS = NP / 8 labels per block.
//   step A: lane (r, c) holds inputs  x[BS c + b]  (replicated over r) and the block  EA[a][b] = E[BS r + a][BS c + b];
//           partial[a] = sum_b EA[a][b] x[b];  all-reduce over c = lane bits 0..2 (three DPP adds per value)
//           -> every lane (r, *) holds the outputs  s[BS r + a]
//   step B: the outputs of A are the inputs: lane (r, c) holds x[BS r + b], block EB[a][b] = E[BS c + a][BS r + b];
//           all-reduce over r = lane bits 3..5 (row_ror:8, v_permlane16_swap, v_permlane32_swap)
//           -> every lane (*, c) holds s[BS c + a]: the input layout of step A again.
// No LDS round trip on the dependent path; 25 FMAs per lane and step at N = 40 (all 64 lanes busy).
// build: hipcc -O3 --offload-arch=gfx950 blocked_matvec.hip -o blocked_matvec ; run: ./blocked_matvec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#ifndef NP
#define NP 40
#endif
constexpr int BS = NP / 8;

__device__ __forceinline__ void swap32(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kXor1 = 0xB1, kXor2 = 0x4E, kHalfMirror = 0x141, kRor8 = 0x128;

template <int K>
__device__ __forceinline__ void partial(const float (&E)[K][K], const float (&x)[K], float (&acc)[K]) {
#pragma unroll
    for (int a = 0; a < K; ++a) acc[a] = E[a][0] * x[0];
#pragma unroll
    for (int b = 1; b < K; ++b)
#pragma unroll
        for (int a = 0; a < K; ++a) acc[a] = fmaf(E[a][b], x[b], acc[a]);
}
template <int K>
__device__ __forceinline__ void reduce_c(float (&v)[K]) {      // all-reduce over lane bits 0..2
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] += dpp<kXor1>(v[a]);
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] += dpp<kXor2>(v[a]);
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] += dpp<kHalfMirror>(v[a]);
}
template <int K>
__device__ __forceinline__ void reduce_r_plain(float (&v)[K]) {      // all-reduce over lane bits 3..5, value by value
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] += dpp<kRor8>(v[a]);
#pragma unroll
    for (int a = 0; a < K; ++a) { float t = v[a]; swap16(v[a], t); v[a] += t; }
#pragma unroll
    for (int a = 0; a < K; ++a) { float t = v[a]; swap32(v[a], t); v[a] += t; }
}
// four values at a time: reduce-scatter over bits 5, 4 (3 swaps + 3 adds), then all-gather (3 copies + 3 swaps)
__device__ __forceinline__ void allreduce4_rows(float &x0, float &x1, float &x2, float &x3) {
    swap32(x0, x1); float z01 = x0 + x1;           // lower half: X0, upper half: X1 (partial over bit 5)
    swap32(x2, x3); float z23 = x2 + x3;
    swap16(z01, z23); float w = z01 + z23;         // rows 0..3 hold four different totals
    float t = w; swap16(w, t);                     // w = [r0 r0 r2 r2], t = [r1 r1 r3 r3]
    float w2 = w, t2 = t;
    swap32(w, w2);                                 // w = [r0 r0 r0 r0], w2 = [r2 ...]
    swap32(t, t2);
    x0 = w; x1 = w2; x2 = t; x3 = t2;              // (which total lands where is fixed; see the check in main)
}
template <int K>
__device__ __forceinline__ void reduce_r_paired(float (&v)[K]) {
#pragma unroll
    for (int a = 0; a < K; ++a) v[a] += dpp<kRor8>(v[a]);
    static_assert(K >= 4 && K <= 8, "");
    allreduce4_rows(v[0], v[1], v[2], v[3]);
    if constexpr (K == 8) allreduce4_rows(v[4], v[5], v[6], v[7]);
    else {
#pragma unroll
        for (int a = 4; a < K; ++a) { float t = v[a]; swap16(v[a], t); v[a] += t; t = v[a]; swap32(v[a], t); v[a] += t; }
    }
}

// MODE 0: LDS broadcast (shipped kernel's step)   1: blocked, plain swaps   2: blocked, paired swaps
// EXTRA: emission factor from an LDS ring (2 reads / step) and row sums exported to an LDS ring (8 lanes, b128 + b32)
template <int MODE, bool EXTRA>
__global__ void __launch_bounds__(64, 1) k(const float *E, const float *v0, float *out, long long *clk, int iters, int N) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    __shared__ __attribute__((aligned(16))) float ering[64][64];     // [slot][8 groups][8]
    __shared__ __attribute__((aligned(16))) float sring[64][64];
    const int lane = threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    typedef float V2 __attribute__((ext_vector_type(2)));
    typedef float V4 __attribute__((ext_vector_type(4)));
    for (int q = lane; q < 64 * 64; q += 64) { (&ering[0][0])[q] = 0.5f; (&sring[0][0])[q] = 0.f; }
    __syncthreads();
    long long t0, t1;
    if constexpr (MODE == 0) {
        V2 e2[NP / 2];
        for (int j = 0; j < NP / 2; ++j) e2[j] = V2{lane < N ? E[lane * N + 2 * j] : 0.f, lane < N ? E[lane * N + 2 * j + 1] : 0.f};
        float p = lane < N ? v0[lane] : 0.f;
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
            lds[lane] = p;
            __builtin_amdgcn_wave_barrier();
            V4 pv[NP / 4];
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) pv[j] = *reinterpret_cast<const V4 *>(lds + 4 * j);
            float ef = 0.5f;
            if (EXTRA) { ef = ering[it & 63][lane]; sring[it & 63][lane] = p; }
            __builtin_amdgcn_sched_barrier(0);
            V2 a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) {
                a0 = __builtin_elementwise_fma(e2[2 * j], pv[j].xy, a0);
                a1 = __builtin_elementwise_fma(e2[2 * j + 1], pv[j].zw, a1);
            }
            V2 a = a0 + a1;
            p = (a.x + a.y) * ef;
            __builtin_amdgcn_wave_barrier();
        }
        t1 = clock64();
        out[lane] = p;
    } else {
        float EA[BS][BS], EB[BS][BS], x[BS];
        for (int a = 0; a < BS; ++a)
            for (int b = 0; b < BS; ++b) {
                const int ia = BS * r + a, ja = BS * c + b;
                EA[a][b] = (ia < N && ja < N) ? E[ia * N + ja] : 0.f;
                const int ib = BS * c + a, jb = BS * r + b;
                EB[a][b] = (ib < N && jb < N) ? E[ib * N + jb] : 0.f;
            }
        for (int b = 0; b < BS; ++b) x[b] = (BS * c + b < N) ? v0[BS * c + b] : 0.f;
        const bool wrA = c == 0, wrB = r == 0;
        t0 = clock64();
        for (int it = 0; it < iters; it += 2) {
            // ---- step A
            {
                float ef[BS];
                if (EXTRA) {
                    const float *ep = &ering[it & 63][8 * r];
#pragma unroll
                    for (int a = 0; a < BS; ++a) ef[a] = ep[a];
                } else {
#pragma unroll
                    for (int a = 0; a < BS; ++a) ef[a] = 0.5f;
                }
                float acc[BS];
                partial<BS>(EA, x, acc);
                reduce_c<BS>(acc);
                if (EXTRA && wrA) {
                    float *sp = &sring[it & 63][8 * r];
#pragma unroll
                    for (int a = 0; a < BS; ++a) sp[a] = acc[a];
                }
#pragma unroll
                for (int a = 0; a < BS; ++a) x[a] = acc[a] * ef[a];
            }
            // ---- step B
            {
                float ef[BS];
                if (EXTRA) {
                    const float *ep = &ering[(it + 1) & 63][8 * c];
#pragma unroll
                    for (int a = 0; a < BS; ++a) ef[a] = ep[a];
                } else {
#pragma unroll
                    for (int a = 0; a < BS; ++a) ef[a] = 0.5f;
                }
                float acc[BS];
                partial<BS>(EB, x, acc);
                if (MODE == 1) reduce_r_plain<BS>(acc); else reduce_r_paired<BS>(acc);
                if (EXTRA && wrB) {
                    float *sp = &sring[(it + 1) & 63][8 * c];
#pragma unroll
                    for (int a = 0; a < BS; ++a) sp[a] = acc[a];
                }
#pragma unroll
                for (int a = 0; a < BS; ++a) x[a] = acc[a] * ef[a];
            }
        }
        t1 = clock64();
        // x is indexed by c again
        if (r == 0) for (int b = 0; b < BS; ++b) out[BS * c + b] = x[b];
    }
    if (lane == 0) clk[0] = t1 - t0;
    if (EXTRA && lane == 1) clk[1] = (long long) sring[3][5];
}

template <int MODE, bool EXTRA>
void run(const float *dE, const float *dv, float *dout, long long *dclk, const std::vector<float> &E, const std::vector<float> &v, int N) {
    for (int iters : {4, 4000}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE, EXTRA>), dim3(1), dim3(64), 0, 0, dE, dv, dout, dclk, iters, N);
        hipDeviceSynchronize();
        float out[64]; long long clk;
        hipMemcpy(out, dout, 64 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
        if (iters == 4) {
            std::vector<double> p(v.begin(), v.end()), s(N);
            for (int it = 0; it < iters; ++it) {
                for (int i = 0; i < N; ++i) { double a = 0; for (int j = 0; j < N; ++j) a += (double) E[i * N + j] * p[j]; s[i] = a; }
                for (int i = 0; i < N; ++i) p[i] = s[i] * 0.5;
            }
            double worst = 0;
            for (int i = 0; i < N; ++i) worst = fmax(worst, fabs(out[i] - p[i]) / fabs(p[i]));
            printf("mode %d extra %d: worst relative error after 4 steps %.2e\n", MODE, (int) EXTRA, worst);
        } else {
            printf("mode %d extra %d: %.1f cycles/step\n", MODE, (int) EXTRA, (double) clk / iters);
        }
    }
}

int main() {
    const int N = NP;
    std::vector<float> E(N * N), v(N);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) E[i * N + j] = 0.02f + 0.03f * ((i * 7 + j * 13) % 11) / 11.f;
        v[i] = 1.0f + 0.01f * i;
    }
    float *dE, *dv, *dout; long long *dclk;
    hipMalloc(&dE, N * N * 4); hipMalloc(&dv, N * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&dclk, 16);
    hipMemcpy(dE, E.data(), N * N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv, v.data(), N * 4, hipMemcpyHostToDevice);
    run<0, false>(dE, dv, dout, dclk, E, v, N);
    run<0, true>(dE, dv, dout, dclk, E, v, N);
    run<1, false>(dE, dv, dout, dclk, E, v, N);
    run<1, true>(dE, dv, dout, dclk, E, v, N);
    run<2, false>(dE, dv, dout, dclk, E, v, N);
    run<2, true>(dE, dv, dout, dclk, E, v, N);
    return 0;
}
