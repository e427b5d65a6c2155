// Developer micro-benchmark (gfx950): cost of the LDS broadcast reads of the per-utterance mat-vec when several
// wavefronts of a compute unit do them at once, with all 64 lanes active vs only the first ACTIVE lanes.
//   each wavefront: ds_write_b32 + 10 x ds_read_b128 (same address in every lane) + 20 v_pk_fma_f32 per step
// build: hipcc -O3 --offload-arch=gfx950 lds_bcast_lanes.hip -o lds_bcast_lanes ; run: ./lds_bcast_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float V2 __attribute__((ext_vector_type(2)));
typedef float V4 __attribute__((ext_vector_type(4)));

template <int ACTIVE>
__global__ void __launch_bounds__(64) k(float *out, long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    const int lane = threadIdx.x;
    V2 e2[20];
    for (int j = 0; j < 20; ++j) e2[j] = V2{0.01f * (lane + j), 0.02f * (j + 1)};
    float p = 1.0f + 0.001f * lane;
    if (lane >= ACTIVE) { out[blockIdx.x * 64 + lane] = 0; return; }       // these lanes are gone for the whole loop
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        lds[lane] = p;
        __builtin_amdgcn_wave_barrier();
        V4 pv[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) pv[j] = *reinterpret_cast<const V4 *>(lds + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
        V2 a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            a0 = __builtin_elementwise_fma(e2[2 * j], V2{pv[j].x, pv[j].y}, a0);
            a1 = __builtin_elementwise_fma(e2[2 * j + 1], V2{pv[j].z, pv[j].w}, a1);
        }
        V2 a = a0 + a1;
        p = (a.x + a.y) * 0.01f + 0.5f;
        __builtin_amdgcn_wave_barrier();
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = p;
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 4000;
    float *dout; long long *dclk;
    hipMalloc(&dout, 256 * 16 * 64 * 4); hipMalloc(&dclk, 256 * 16 * 8);
    for (int per_cu : {1, 4, 8}) {
        const int grid = 256 * per_cu;
        for (int act : {64, 48, 41, 32}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (act == 64) hipLaunchKernelGGL(k<64>, dim3(grid), dim3(64), 0, 0, dout, dclk, iters);
                if (act == 48) hipLaunchKernelGGL(k<48>, dim3(grid), dim3(64), 0, 0, dout, dclk, iters);
                if (act == 41) hipLaunchKernelGGL(k<41>, dim3(grid), dim3(64), 0, 0, dout, dclk, iters);
                if (act == 32) hipLaunchKernelGGL(k<32>, dim3(grid), dim3(64), 0, 0, dout, dclk, iters);
            }
            hipDeviceSynchronize();
            static long long h[256 * 16];
            hipMemcpy(h, dclk, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; long long mx = 0;
            for (int i = 0; i < grid; ++i) { s += h[i]; mx = h[i] > mx ? h[i] : mx; }
            printf("%d wavefront(s) per CU, %2d active lanes: mean %.1f, max %.1f cycles/step\n", per_cu, act, s / grid / iters, (double) mx / iters);
        }
    }
    return 0;
}
