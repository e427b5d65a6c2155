// Developer micro-benchmark (gfx950): cost of the LDS broadcast reads of the per-utterance mat-vec when several
// wavefronts of a compute unit do them at once, with all 64 lanes active vs only the first ACTIVE lanes.
//   each wavefront: ds_write_b32 + 10 x ds_read_b128 (same address in every lane) + 20 v_pk_fma_f32 per step
// build: hipcc -O3 --offload-arch=gfx950 lds_bcast_lanes.hip -o lds_bcast_lanes ; run: ./lds_bcast_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float V2 __attribute__((ext_vector_type(2)));
typedef float V4 __attribute__((ext_vector_type(4)));

template <int ACTIVE, int FLAGS>
__global__ void __launch_bounds__(64, 1) k(float *out, long long *clk, int iters, float *sink) {
    __shared__ __attribute__((aligned(16))) float lds[64];
    const int lane = threadIdx.x;
    V2 e2[20];
    for (int j = 0; j < 20; ++j) e2[j] = V2{0.01f * (lane + j), 0.02f * (j + 1)};
    float p = 1.0f + 0.001f * lane, sprev = 1.0f, lg = 0.f, arg = 0.2f, ee = 1.0f; unsigned wlo = ~0u, whi = 0u;
    if (lane >= ACTIVE) { out[blockIdx.x * 64 + lane] = 0; return; }       // these lanes are gone for the whole loop
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        lds[lane] = p;
        __builtin_amdgcn_wave_barrier();
        V4 pv[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) pv[j] = *reinterpret_cast<const V4 *>(lds + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
        if (FLAGS & 1) { lg = __builtin_amdgcn_logf(sprev) + arg; arg = ee * 0.3f + 0.1f; ee = __builtin_amdgcn_exp2f(arg - 1.0f); }
        if (FLAGS & 2) sink[(size_t) blockIdx.x * 64 * 4096 + (size_t) (it & 4095) * 64 + lane] = lg;
        if (FLAGS & 4) { unsigned sb = __float_as_uint(sprev); wlo = min(wlo, sb); whi = max(whi, sb); }
        if ((FLAGS & 8) && (it & 3) == 2) { int ex = __builtin_amdgcn_readlane(__builtin_amdgcn_frexp_expf(sprev), 40); arg -= (float) ex; ee = __builtin_amdgcn_exp2f(arg - 1.0f); }
        __builtin_amdgcn_sched_barrier(0);
        V2 a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            a0 = __builtin_elementwise_fma(e2[2 * j], V2{pv[j].x, pv[j].y}, a0);
            a1 = __builtin_elementwise_fma(e2[2 * j + 1], V2{pv[j].z, pv[j].w}, a1);
        }
        V2 a = a0 + a1;
        sprev = (a.x + a.y) * 0.01f + 0.5f;
        p = sprev * ee;
        __builtin_amdgcn_wave_barrier();
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = p + lg + (float) (wlo ^ whi);
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 4000;
    float *dout; long long *dclk;
    hipMalloc(&dout, 256 * 16 * 64 * 4); hipMalloc(&dclk, 256 * 16 * 8);
    float *sink; hipMalloc(&sink, (size_t) 256 * 8 * 64 * 4096 * 4);
    for (int per_cu : {1, 4, 8}) {
        const int grid = 256 * per_cu;
        for (int fl : {0, 1, 2, 3, 4, 8, 15}) {
            for (int rep = 0; rep < 2; ++rep) {
#define L(F) if (fl == F) hipLaunchKernelGGL((k<64, F>), dim3(grid), dim3(64), 0, 0, dout, dclk, iters, sink);
                L(0) L(1) L(2) L(3) L(4) L(8) L(15)
            }
            hipDeviceSynchronize();
            static long long h[256 * 16];
            hipMemcpy(h, dclk, grid * 8, hipMemcpyDeviceToHost);
            double sm = 0; long long mx = 0;
            for (int i = 0; i < grid; ++i) { sm += h[i]; mx = h[i] > mx ? h[i] : mx; }
            printf("%d wavefront(s) per CU, extras %2d (1 exp+log, 2 store, 4 watch, 8 rescale): mean %.1f, max %.1f cycles/step\n", per_cu, fl, sm / grid / iters, (double) mx / iters);
        }
    }
    return 0;
}
