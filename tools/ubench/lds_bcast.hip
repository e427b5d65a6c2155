// Developer micro-benchmark (gfx950): what does one "broadcast a 40-vector through LDS + 40x40 mat-vec" step cost,
// and what clock does a single latency-bound wave per CU actually run at?
// build: hipcc -O3 --offload-arch=gfx950 lds_bcast.hip -o lds_bcast ; run: ./lds_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float V2 __attribute__((ext_vector_type(2)));
typedef float V4 __attribute__((ext_vector_type(4)));
constexpr int NP = 40;

template <int SH, int PLACE, int TR>
__global__ void __launch_bounds__(64, 1) ksh(float *out, long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    const int lane = threadIdx.x;
    V2 e2[NP / 2];
    for (int j = 0; j < NP / 2; ++j) e2[j] = V2{1.0f / NP + 1e-4f * (lane + j), 1.0f / NP - 1e-4f * j};
    float p = 1.0f + lane * 1e-3f;
    float d[8];
    for (int q = 0; q < 8; ++q) d[q] = 1.0f + q + lane;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        lds[lane] = p;
        __builtin_amdgcn_wave_barrier();
        V4 pv[NP / 4];
        if (PLACE == 1) {
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) {
                pv[j] = *reinterpret_cast<const V4 *>(lds + 4 * j);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < (SH + NP / 4 - 1) / (NP / 4); ++q)
                    if (j * ((SH + NP / 4 - 1) / (NP / 4)) + q < SH) d[(j + q) & 7] = fmaf(d[(j + q) & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) pv[j] = *reinterpret_cast<const V4 *>(lds + 4 * j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < SH; ++q) d[q & 7] = fmaf(d[q & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int q = 0; q < TR; ++q) d[q & 7] = (q & 1) ? __builtin_amdgcn_logf(d[q & 7]) : __builtin_amdgcn_exp2f(d[q & 7]);
            __builtin_amdgcn_sched_barrier(0);
        }
        V2 a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) {
            a0 = __builtin_elementwise_fma(e2[2 * j], pv[j].xy, a0);
            a1 = __builtin_elementwise_fma(e2[2 * j + 1], pv[j].zw, a1);
        }
        V2 a = a0 + a1;
        float s = a.x + a.y;
        __builtin_amdgcn_wave_barrier();
        p = s * 0.999f;
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float acc = p;
    for (int q = 0; q < 8; ++q) acc += d[q];
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int SH, int PLACE, int TR> void runsh(const char *name) {
    float *out; long long *clk;
    hipMalloc(&out, 64 * 4); hipMalloc(&clk, 16);
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((ksh<SH, PLACE, TR>), dim3(1), dim3(64), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-40s %.1f cycles/iter\n", name, (double) h[0] / iters);
    hipFree(out); hipFree(clk);
}

template <int MODE>
__global__ void __launch_bounds__(64, 1) k(float *out, long long *clk, int iters, int nact) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    const int lane = threadIdx.x;
    V2 e2[NP / 2];
    for (int j = 0; j < NP / 2; ++j) e2[j] = V2{1.0f / NP + 1e-4f * (lane + j), 1.0f / NP - 1e-4f * j};
    float p = 1.0f + lane * 1e-3f;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5 && lane >= nact) continue;
        lds[lane] = p;
        __builtin_amdgcn_wave_barrier();
        V4 pv[NP / 4];
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) pv[j] = *reinterpret_cast<const V4 *>(lds + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
        float s;
        if (MODE == 0) {            // reads only: consume with one add per read
            float a = 0;
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) a += pv[j].x;
            s = a;
        } else if (MODE == 1 || MODE == 5) {     // 2 chains of pk_fma
            V2 a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) {
                a0 = __builtin_elementwise_fma(e2[2 * j], pv[j].xy, a0);
                a1 = __builtin_elementwise_fma(e2[2 * j + 1], pv[j].zw, a1);
            }
            V2 a = a0 + a1;
            s = a.x + a.y;
        } else if (MODE == 2) {     // 4 chains
            V2 a0 = {0, 0}, a1 = {0, 0}, a2 = {0, 0}, a3 = {0, 0};
#pragma unroll
            for (int j = 0; j + 1 < NP / 4; j += 2) {
                a0 = __builtin_elementwise_fma(e2[2 * j], pv[j].xy, a0);
                a1 = __builtin_elementwise_fma(e2[2 * j + 1], pv[j].zw, a1);
                a2 = __builtin_elementwise_fma(e2[2 * j + 2], pv[j + 1].xy, a2);
                a3 = __builtin_elementwise_fma(e2[2 * j + 3], pv[j + 1].zw, a3);
            }
            V2 a = (a0 + a1) + (a2 + a3);
            s = a.x + a.y;
        } else if (MODE == 3) {     // scalar fma, 4 chains
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int j = 0; j < NP / 4; ++j) {
                a0 = fmaf(e2[2 * j].x, pv[j].x, a0);
                a1 = fmaf(e2[2 * j].y, pv[j].y, a1);
                a2 = fmaf(e2[2 * j + 1].x, pv[j].z, a2);
                a3 = fmaf(e2[2 * j + 1].y, pv[j].w, a3);
            }
            s = (a0 + a1) + (a2 + a3);
        } else {                    // MODE 4: no LDS at all: readlane broadcast
            V2 a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
            for (int j = 0; j < NP; j += 4) {
                V2 v0 = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), j)),
                         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), j + 1))};
                V2 v1 = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), j + 2)),
                         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), j + 3))};
                a0 = __builtin_elementwise_fma(e2[j / 2], v0, a0);
                a1 = __builtin_elementwise_fma(e2[j / 2 + 1], v1, a1);
            }
            V2 a = a0 + a1;
            s = a.x + a.y;
        }
        __builtin_amdgcn_wave_barrier();
        p = s * 0.999f;
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 64 + lane] = p;
    if (lane == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE> void run(const char *name, int grid, int nact = 64) {
    float *out; long long *clk;
    hipMalloc(&out, grid * 64 * 4); hipMalloc(&clk, grid * 16);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, clk, iters, nact);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * grid);
    hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    printf("%-28s grid %4d: %.1f ns/iter (event), clock64 %.1f /iter, wall_clock64 %.2f ticks/iter -> shader clock ~ %.0f MHz if wall=100MHz\n",
           name, grid, ms * 1e6 / iters, (double) h[0] / iters, (double) h[1] / iters, (double) h[0] / (double) h[1] * 100.0);
    hipFree(out); hipFree(clk);
}

int main() {
    runsh<0, 0, 0>("shadow 0");
    runsh<4, 0, 0>("shadow 4 valu after reads");
    runsh<8, 0, 0>("shadow 8 valu after reads");
    runsh<12, 0, 0>("shadow 12 valu after reads");
    runsh<16, 0, 0>("shadow 16 valu after reads");
    runsh<24, 0, 0>("shadow 24 valu after reads");
    runsh<32, 0, 0>("shadow 32 valu after reads");
    runsh<0, 0, 2>("shadow 2 transcendental after reads");
    runsh<6, 0, 2>("shadow 6 valu + 2 transc after reads");
    runsh<10, 1, 0>("shadow 10 valu interleaved w/ reads");
    runsh<20, 1, 0>("shadow 20 valu interleaved w/ reads");
    runsh<30, 1, 0>("shadow 30 valu interleaved w/ reads");
    for (int grid : {1, 256}) {
        run<0>("reads only", grid);
        run<1>("pk_fma 2 chains", grid);
        run<2>("pk_fma 4 chains", grid);
        run<3>("fma 4 chains", grid);
        run<4>("readlane, no LDS", grid);
        run<5>("pk_fma 2 chains, 41 lanes", grid, 41);
    }
    return 0;
}
