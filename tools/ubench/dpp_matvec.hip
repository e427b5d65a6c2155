// Developer micro-benchmark (gfx950): the 40x40 mat-vec of one recursion step WITHOUT the LDS broadcast.
//   lane l = 16 r + c holds label 4 c + r  (r = row of 16 lanes, c < NP/4)
//   acc_k(r, c) = sum_m E[4c+k][4m+r] * v(lane 16 r + m)         40 v_fmac_f32_dpp row_newbcast:m
//   s[4c+k] = sum_r acc_k(r, c), delivered to row k              reduce-scatter: 3 permlane swaps + 3 adds
// build: hipcc -O3 --offload-arch=gfx950 dpp_matvec.hip -o dpp_matvec ; run: ./dpp_matvec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int NP = 40;
constexpr int NC = NP / 4;

__device__ __forceinline__ void swap_halves(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_rows(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

template <int M>
__device__ __forceinline__ void fmac_bcast(float &acc, float v, float e) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(e), "n"(M));
}

#ifndef NA
#define NA 1
#endif
template <int M>
__device__ __forceinline__ void col_step(float (&acc)[4 * NA], float v, const float (&e)[4][NC]) {
    if constexpr (M < NC) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fmac_bcast<M>(acc[k + 4 * (M % NA)], v, e[k][M]);
        col_step<M + 1>(acc, v, e);
    }
}

__device__ __forceinline__ float dpp_matvec(float v, const float (&e)[4][NC]) {
    float acc[4 * NA];
#pragma unroll
    for (int k = 0; k < 4 * NA; ++k) acc[k] = 0;
    asm volatile("s_nop 1" ::: "memory");       // VALU write of v -> DPP read: 2 wait states
    col_step<0>(acc, v, e);
#pragma unroll
    for (int a = 1; a < NA; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += acc[k + 4 * a];
    swap_halves(acc[0], acc[2]);
    swap_halves(acc[1], acc[3]);
    float X = acc[0] + acc[2], Y = acc[1] + acc[3];
    swap_rows(X, Y);
    return X + Y;
}

// MODE 0: DPP mat-vec;  MODE 1: the LDS-broadcast mat-vec of the shipped kernel (for comparison in the same run)
template <int MODE>
__global__ void __launch_bounds__(64, 1) k(const float *E, const float *v0, float *out, long long *clk, int iters, int N) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    const int lane = threadIdx.x;
    const int r = lane >> 4, c = lane & 15;
    float e[4][NC];
    typedef float V2 __attribute__((ext_vector_type(2)));
    typedef float V4 __attribute__((ext_vector_type(4)));
    V2 e2[NP / 2];
    float p;
    if (MODE == 0) {
        for (int kk = 0; kk < 4; ++kk)
            for (int m = 0; m < NC; ++m) {
                const int i = 4 * c + kk, j = 4 * m + r;
                e[kk][m] = (c < NC && i < N && j < N) ? E[i * N + j] : 0.f;
            }
        const int lab = 4 * c + r;
        p = (c < NC && lab < N) ? v0[lab] : 0.f;
    } else {
        for (int j = 0; j < NP / 2; ++j) e2[j] = V2{lane < N ? E[lane * N + 2 * j] : 0.f, lane < N ? E[lane * N + 2 * j + 1] : 0.f};
        p = lane < N ? v0[lane] : 0.f;
    }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float s;
        if (MODE == 0) {
            s = dpp_matvec(p, e);
        } else {
            lds[lane] = p;
            __builtin_amdgcn_wave_barrier();
            V4 pv[NP / 4];
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) pv[j] = *reinterpret_cast<const V4 *>(lds + 4 * j);
            __builtin_amdgcn_sched_barrier(0);
            V2 a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) {
                a0 = __builtin_elementwise_fma(e2[2 * j], pv[j].xy, a0);
                a1 = __builtin_elementwise_fma(e2[2 * j + 1], pv[j].zw, a1);
            }
            V2 a = a0 + a1;
            s = a.x + a.y;
            __builtin_amdgcn_wave_barrier();
        }
        p = s * 0.5f;
    }
    long long t1 = clock64();
    out[lane] = p;
    if (lane == 0) clk[0] = t1 - t0;
}

int main() {
    const int N = 40;
    std::vector<float> E(N * N), v(N);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) E[i * N + j] = 0.02f + 0.03f * ((i * 7 + j * 13) % 11) / 11.f;
        v[i] = 1.0f + 0.01f * i;
    }
    float *dE, *dv, *dout; long long *dclk;
    hipMalloc(&dE, N * N * 4); hipMalloc(&dv, N * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&dclk, 8);
    hipMemcpy(dE, E.data(), N * N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv, v.data(), N * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        for (int iters : {3, 4000}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dE, dv, dout, dclk, iters, N);
                else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dE, dv, dout, dclk, iters, N);
            }
            hipDeviceSynchronize();
            float out[64]; long long clk;
            hipMemcpy(out, dout, 64 * 4, hipMemcpyDeviceToHost);
            hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
            if (iters == 3) {
                std::vector<double> p(v.begin(), v.end()), s(N);
                for (int it = 0; it < iters; ++it) {
                    for (int i = 0; i < N; ++i) { double a = 0; for (int j = 0; j < N; ++j) a += (double) E[i * N + j] * p[j]; s[i] = a; }
                    for (int i = 0; i < N; ++i) p[i] = s[i] * 0.5;
                }
                double worst = 0;
                for (int i = 0; i < N; ++i) {
                    const int lane = mode == 0 ? 16 * (i % 4) + i / 4 : i;
                    worst = fmax(worst, fabs(out[lane] - p[i]) / fabs(p[i]));
                }
                printf("mode %d: worst relative error after 3 steps %.2e\n", mode, worst);
            } else {
                printf("mode %d: %.1f cycles/step (%s)\n", mode, (double) clk / iters, mode == 0 ? "DPP row_newbcast + permlane reduce-scatter" : "LDS broadcast");
            }
        }
    }
    return 0;
}
