// Developer micro-benchmark (gfx950): fp32 MFMA rate when the A operands come from a stream of global loads (as in the
// large-alphabet step) or from LDS, instead of from registers that never change (mfma_issue.hip: 32 cycles per MFMA).
// build: hipcc -O3 --offload-arch=gfx950 mfma_fed.hip -o mfma_fed ; run: ./mfma_fed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float V4 __attribute__((ext_vector_type(4)));

// MODE 0: operands in registers (16 distinct A, 8 distinct B);  1: A from global loads two iterations ahead (L2-resident
// buffer);  2: A from global, HBM stream (non-temporal, large buffer);  3: A from LDS reads one iteration ahead
template <int MODE>
__global__ void __launch_bounds__(256) k(const V4 *src, size_t nv4, float *out, long long *clk, int iters) {
    __shared__ V4 lds[4][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float b[8];
    for (int j = 0; j < 8; ++j) b[j] = 0.5f - 1e-3f * (lane - j);
    V4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = V4{0, 0, 0, 0};
    const size_t stride = (size_t) gridDim.x * 256 * 4;
    const V4 *p = src + ((size_t) blockIdx.x * 256 + threadIdx.x) * 4;
    V4 x0[4], x1[4], x2[4];
    for (int q = 0; q < 4; ++q) { x0[q] = p[q]; x1[q] = p[q]; x2[q] = p[q]; lds[wave][lane * 4 + q] = p[q]; }
    __syncthreads();
    size_t off = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        V4 cur[4];
        for (int q = 0; q < 4; ++q) cur[q] = x0[q];
        if (MODE == 1 || MODE == 2) {
            for (int q = 0; q < 4; ++q) { x0[q] = x1[q]; x1[q] = x2[q]; }
            off += stride;
            if (off + stride > nv4) off = 0;
            for (int q = 0; q < 4; ++q) x2[q] = MODE == 2 ? __builtin_nontemporal_load(p + off + q) : p[(off & 0xffff) + q];
        } else if (MODE == 3) {
            for (int q = 0; q < 4; ++q) x0[q] = lds[wave][((lane + it) & 63) * 4 + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[2 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].x, b[0], c[2 * q], 0, 0, 0);
            c[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].x, b[1], c[2 * q + 1], 0, 0, 0);
            c[2 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].y, b[2], c[2 * q], 0, 0, 0);
            c[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].y, b[3], c[2 * q + 1], 0, 0, 0);
            c[2 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].z, b[4], c[2 * q], 0, 0, 0);
            c[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].z, b[5], c[2 * q + 1], 0, 0, 0);
            c[2 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].w, b[6], c[2 * q], 0, 0, 0);
            c[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q].w, b[7], c[2 * q + 1], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run(const char *what, const V4 *src, size_t nv4) {
    float *out; long long *clk;
    (void) hipMalloc(&out, 256 * 256 * 4); (void) hipMalloc(&clk, 8);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, src, nv4, out, clk, 10);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, src, nv4, out, clk, iters);
    (void) hipEventRecord(e1, 0);
    (void) hipDeviceSynchronize();
    float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
    long long h = 0; (void) hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * 4 * iters * 32.0 * 2048.0, bytes = MODE == 1 || MODE == 2 ? 256.0 * 256 * 64.0 * iters : 0;
    printf("%-52s %6.1f cycles per MFMA (wavefront 0); kernel %.3f ms = %.0f TFLOP/s, %.2f TB/s of operands\n", what, (double) h / (iters * 32.0), ms,
           flops / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e12);
    (void) hipFree(out); (void) hipFree(clk);
}

int main() {
    const size_t nv4 = (size_t) 1 << 26;             // 1 GiB of float4
    V4 *src; (void) hipMalloc(&src, nv4 * 16); (void) hipMemset(src, 0, nv4 * 16);
    run<0>("A operands in registers", src, nv4);
    run<1>("A operands from global loads (cache-resident)", src, nv4);
    run<2>("A operands from global loads (HBM stream, nt)", src, nv4);
    run<3>("A operands from LDS", src, nv4);
    return 0;
}
