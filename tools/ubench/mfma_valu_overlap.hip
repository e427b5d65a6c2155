// Developer micro-benchmark (gfx950): does v_mfma_f32_16x16x4_f32 run BESIDE vector-ALU instructions, or instead of them?
// Per loop trip: 8 matrix instructions (4 accumulators: no dependency stalls) and/or NV independent v_fma_f32, interleaved one
// matrix instruction : NV / 8 fmas (sched_barrier pins the order).  Timed with events over the whole kernel, 1 / 2 / 4
// wavefronts per SIMD, every CU busy.  If the matrix pipe were separate, "both" would cost max(mfma, valu); if the fp32 matrix
// instruction occupies the vector ALU, it costs the sum.  A second pair of kernels puts the matrix instructions and the fmas
// into DIFFERENT wavefronts of one SIMD (even / odd wavefronts).
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float V4 __attribute__((ext_vector_type(4)));

// MODE 1: matrix only, 2: fma only, 3: both interleaved in one wavefront, 4: even wavefronts matrix / odd wavefronts fma
template <int MODE, int NV>
__global__ void __launch_bounds__(1024) k(float *out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[8], b[8], f[16];
    for (int j = 0; j < 8; ++j) { a[j] = 1.0f + 1e-3f * (lane + j); b[j] = 0.5f - 1e-3f * (lane - j); }
    for (int j = 0; j < 16; ++j) f[j] = 1.0f + 1e-4f * (lane + j);
    V4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = V4{0, 0, 0, 0};
    const float m = 0.999f, ad = 1e-3f;
    const bool do_m = MODE == 1 || MODE == 3 || (MODE == 4 && (wave & 4) == 0);      // (waves w and w + 4 share a SIMD)
    const bool do_v = MODE == 2 || MODE == 3 || (MODE == 4 && (wave & 4) != 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (do_m) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c[j & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (do_v) {
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) { const int i = (j * (NV / 8) + q) & 15; f[i] = __builtin_fmaf(f[i], m, ad); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][3];
    for (int j = 0; j < 16; ++j) s += f[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
double run(int threads) {
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e6 / iters;        // ns per loop trip
}

template <int NV> void table() {
    for (int threads : {256, 512, 1024}) {
        const double tm = run<1, NV>(threads), tv = run<2, NV>(threads), tb = run<3, NV>(threads);
        printf("%2d fma per 8 mfma, %d wavefront(s)/SIMD: mfma only %6.1f ns, fma only %6.1f ns, both in one wavefront %6.1f ns  (max %6.1f, sum %6.1f)\n",
               NV, threads / 256, tm, tv, tb, tm > tv ? tm : tv, tm + tv);
    }
    for (int threads : {512, 1024}) {
        const double t4 = run<4, NV>(threads);
        printf("%2d fma per 8 mfma, %d wavefront(s)/SIMD, half of them matrix-only and half fma-only: %6.1f ns per trip\n", NV, threads / 256, t4);
    }
}

int main() {
    table<32>();
    table<64>();
    table<128>();
    return 0;
}
