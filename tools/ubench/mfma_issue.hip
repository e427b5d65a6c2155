// Developer micro-benchmark (gfx950): how often can ONE wavefront issue an fp32 MFMA, and how many wavefronts per SIMD
// does it take to keep the matrix pipe at its rate?  (fwd_cluster_kernel measured 52 cycles per v_mfma_f32_16x16x4_f32
// with one wavefront per SIMD; the pipe's rate is 32.)
// (round 4: also v_mfma_f32_4x4x1_16B_f32 -- sixteen independent 4 x 4 outer products)
// build: hipcc -O3 --offload-arch=gfx950 mfma_issue.hip -o mfma_issue ; run: ./mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float V4 __attribute__((ext_vector_type(4)));
typedef float V16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ void __launch_bounds__(1024) k(float *out, long long *clk, int iters) {
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = 1.0f + 1e-3f * (lane + j); b[j] = 0.5f - 1e-3f * (lane - j); }
    V4 c4[NACC];
    V16 c16[NACC > 2 ? 2 : NACC];
    for (int j = 0; j < NACC; ++j) c4[j] = V4{0, 0, 0, 0};
    for (int j = 0; j < (NACC > 2 ? 2 : NACC); ++j) for (int q = 0; q < 16; ++q) c16[j][q] = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SHAPE == 0) c4[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c4[j % NACC], 0, 0, 0);
            else if (SHAPE == 2) c4[j % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], c4[j % NACC], 0, 0, 0);
            else c16[j % (NACC > 2 ? 2 : NACC)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c16[j % (NACC > 2 ? 2 : NACC)], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += c4[j][0] + c4[j][3];
    for (int j = 0; j < (NACC > 2 ? 2 : NACC); ++j) s += c16[j][0] + c16[j][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int SHAPE, int NACC>
void run(const char *what, int threads) {
    float *out; long long *clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 8);
    const int iters = 20000;
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, clk, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h = 0; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double per = (double) h / (iters * 8.0);
    const double flops = 256.0 * (threads / 64) * iters * 8.0 * 2048.0 * (SHAPE == 0 ? 1.0 : SHAPE == 2 ? 0.25 : 2.0);
    printf("%-30s %2d wavefront(s) per SIMD: %6.1f ticks per MFMA per wavefront, %5.1f per SIMD; kernel %.3f ms = %.0f TFLOP/s (events)\n", what, threads / 256, per,
           per / (threads / 256), ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<0, 1>("16x16x4 f32, 1 accumulator", threads);
        run<0, 4>("16x16x4 f32, 4 accumulators", threads);
        run<0, 8>("16x16x4 f32, 8 accumulators", threads);
        run<2, 1>("4x4x1 (16 blocks) f32, 1 accumulator", threads);
        run<2, 4>("4x4x1 (16 blocks) f32, 4 accumulators", threads);
        run<2, 8>("4x4x1 (16 blocks) f32, 8 accumulators", threads);
        run<1, 1>("32x32x2 f32, 1 accumulator", threads);
        run<1, 2>("32x32x2 f32, 2 accumulators", threads);
    }
    return 0;
}
