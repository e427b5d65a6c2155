// Developer micro-benchmark (gfx950): issue cost of the pieces of the "three bfloat16 per float" product
// (asg_generic.hip, fwd_step_mfma): v_mfma_f32_16x16x32_bf16, v_cvt_pk_bf16_f32, v_pk_add_f32, v_and / v_lshlrev, alone and
// interleaved, one wavefront per SIMD (256-thread workgroups, one per CU).  Cycles per instruction from s_memtime-free
// clock64() over a long unrolled loop.
// build: hipcc -O3 --offload-arch=gfx950 bf16_split_rates.hip -o bf16_split_rates ; run: ./bf16_split_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 BF8 __attribute__((ext_vector_type(8)));
typedef float V4 __attribute__((ext_vector_type(4)));
typedef float V2 __attribute__((ext_vector_type(2)));
typedef unsigned U4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, long long *clk, int iters) {
    const int lane = threadIdx.x;
    U4 ua = {0x3f803f80u + lane, 0x3f803f81u, 0x3f813f80u, 0x3f803f82u}, ub = {0x3f803f80u, 0x3f823f80u + lane, 0x3f803f80u, 0x3f803f83u};
    BF8 a = __builtin_bit_cast(BF8, ua), b = __builtin_bit_cast(BF8, ub);
    V4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = V4{0, 0, 0, 0};
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = 1.0f + 0.001f * (lane + j);
    unsigned y[16];
    for (int j = 0; j < 16; ++j) y[j] = lane * 77u + j;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 5) {          // 8 independent MFMAs
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 5) {          // 8 cvt_pk
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y[j]) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
        }
        if (MODE == 2 || MODE == 5) {          // 8 pk_add
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                V2 v = {x[2 * j], x[2 * j + 1]};
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(v));
                x[2 * j] = v.x; x[2 * j + 1] = v.y;
            }
        }
        if (MODE == 3 || MODE == 5) {          // 16 and / shift
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j]) : "v"(y[j]));
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(y[j + 8]) : "v"(y[j + 8]));
            }
        }
        if (MODE == 4) {                       // 8 dependent MFMAs on ONE accumulator
#pragma unroll
            for (int j = 0; j < 8; ++j) c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[0], 0, 0, 0);
        }
        if (MODE == 6) {                       // MFMA with 3 valu ops interleaved after each (the kernel's intended shape)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y[j]) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j + 8]) : "v"(y[j + 8]));
                V2 v = {x[2 * j], x[2 * j + 1]};
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(v));
                x[2 * j] = v.x; x[2 * j + 1] = v.y;
            }
        }
        if (MODE == 7) {                       // fp32 MFMA for reference
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], x[j + 8], c[j], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][3];
    for (int j = 0; j < 16; ++j) s += x[j] + (float) y[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run(const char *what, int per_iter) {
    float *out; long long *clk;
    (void) hipMalloc(&out, 256 * 256 * 4); (void) hipMalloc(&clk, 8);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, clk, 10);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, clk, iters);
    (void) hipEventRecord(e1, 0);
    (void) hipDeviceSynchronize();
    float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
    long long h = 0; (void) hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f clock64 ticks / iter, %7.2f ns / iter (events), %6.2f ns per instruction\n", what, (double) h / iters, ms * 1e6 / iters, ms * 1e6 / iters / per_iter);
    (void) hipFree(out); (void) hipFree(clk);
}

int main() {
    run<0>("8 independent v_mfma_f32_16x16x32_bf16", 8);
    run<4>("8 dependent v_mfma_f32_16x16x32_bf16 (one accumulator)", 8);
    run<7>("8 independent v_mfma_f32_16x16x4_f32", 8);
    run<1>("8 v_cvt_pk_bf16_f32", 8);
    run<2>("8 v_pk_add_f32", 8);
    run<3>("8 v_and_b32 + 8 v_lshlrev_b32", 16);
    run<5>("all of the above one after the other (8+8+8+16)", 40);
    run<6>("8 x (mfma, cvt_pk, and, pk_add) interleaved", 32);
    return 0;
}
