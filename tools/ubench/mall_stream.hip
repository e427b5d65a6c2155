// Developer micro-benchmark (gfx950): what does the 256 MiB Infinity Cache do for a stream that is re-read every "frame"?
// The large-alphabet step (asg_generic.hip, fwd_step_mfma) reads the same 800 MB (two 400 MB operand-order copies of the
// transition matrix) once per frame, every workgroup its own contiguous tile, front to back.  A cyclic stream larger than
// an LRU cache never hits; walking the tile BACK to front on odd frames ("serpentine") makes the last bytes of frame t the
// first of frame t + 1.  This probe streams a buffer of a given size with 250 workgroups x 4 wavefronts, each wavefront a
// contiguous quarter of its workgroup's tile, in one of the two orders, with default-policy or non-temporal loads, and
// prints us per frame (HIP events over 40 frames).
// build: hipcc -O3 --offload-arch=gfx950 mall_stream.hip -o mall_stream ; run: ./mall_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float V4 __attribute__((ext_vector_type(4)));

// per wavefront: nv4 float4 "rows" of 64 lanes (1 KB per wavefront load)
template <bool NT>
__global__ void __launch_bounds__(1024) stream_kernel(const V4 *src, size_t rows_per_wave, int reverse, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const V4 *p = src + ((size_t) (blockIdx.x * nw + wave) * rows_per_wave) * 64 + lane;
    V4 acc = {0, 0, 0, 0};
    constexpr int U = 8;
    const size_t n = rows_per_wave / U * U;
    for (size_t r = 0; r < n; r += U) {
        V4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t rr = reverse ? (n - 1 - (r + u)) : (r + u);
            v[u] = NT ? __builtin_nontemporal_load(p + rr * 64) : p[rr * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

static int g_waves = 4;
template <bool NT>
static double run(const V4 *buf, size_t bytes, int serp, float *out, int frames) {
    const int wgs = 250;
    const size_t rows_per_wave = bytes / 1024 / (wgs * g_waves);
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    for (int f = 0; f < 6; ++f) hipLaunchKernelGGL((stream_kernel<NT>), dim3(wgs), dim3(64 * g_waves), 0, 0, buf, rows_per_wave, serp ? (f & 1) : 0, out);
    (void) hipEventRecord(e0, 0);
    for (int f = 0; f < frames; ++f) hipLaunchKernelGGL((stream_kernel<NT>), dim3(wgs), dim3(64 * g_waves), 0, 0, buf, rows_per_wave, serp ? (f & 1) : 0, out);
    (void) hipEventRecord(e1, 0);
    (void) hipEventSynchronize(e1);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / frames;
}

int main() {
    const size_t maxb = (size_t) 1200 << 20;
    V4 *buf; float *out;
    if (hipMalloc(&buf, maxb) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void) hipMalloc(&out, 64);
    (void) hipMemset(buf, 0, maxb);
    (void) hipDeviceSynchronize();
    printf("%8s %6s %6s %10s %8s\n", "MB", "order", "policy", "us/frame", "TB/s");
    const size_t sizes[] = {128, 200, 256, 320, 400, 512, 640, 800, 1000, 1200};
    for (size_t mb : sizes) {
        const size_t bytes = (mb << 20) / (1024 * 1000) * (1024 * 1000);      // whole 1 KB rows for 1000 wavefronts
        for (int serp = 0; serp < 2; ++serp)
            for (int nt = 0; nt < 2; ++nt) {
                const double us = nt ? run<true>(buf, bytes, serp, out, 40) : run<false>(buf, bytes, serp, out, 40);
                printf("%8zu %6s %6s %10.1f %8.2f\n", mb, serp ? "serp" : "cyclic", nt ? "nt" : "dflt", us, bytes / us / 1e6);
            }
    }
    // the same 800 MB with more wavefronts per compute unit: is the per-CU rate a per-wavefront limit?
    printf("waves per CU, 800 MB, cyclic, nt:\n");
    for (int w : {1, 2, 4, 8, 16}) {
        g_waves = w;
        const size_t bytes = ((size_t) 800 << 20) / (1024 * 4000) * (1024 * 4000);
        const double us = run<true>(buf, bytes, 0, out, 40);
        printf("%8d waves %10.1f us %8.2f TB/s\n", w, us, bytes / us / 1e6);
    }
    return 0;
}
