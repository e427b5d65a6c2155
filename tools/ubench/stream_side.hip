// Developer micro-benchmark (gfx950): does L2-resident side traffic slow an HBM stream?  The large-alphabet step reads, per 10 KB of
// its 800 MB matrix stream, 4 KB of the batch's vectors (1.28 MB per direction, re-read by every workgroup: L2 hits).  Here 250
// workgroups x 4 wavefronts stream 800 MB with non-temporal loads (10 x 1 KB per wavefront and round) and additionally read SIDE KB per
// round from a 1.28 MB buffer (default policy).  Prints us per frame for SIDE = 0, 2, 4, 8.
// build: hipcc -O3 --offload-arch=gfx950 stream_side.hip -o stream_side
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float V4 __attribute__((ext_vector_type(4)));
template <int SIDE>
__global__ void __launch_bounds__(256) k(const V4 *src, size_t rounds, const V4 *side, size_t side_rows, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const V4 *p = src + ((size_t) (blockIdx.x * 4 + wave) * rounds * 10) * 64 + lane;
    const V4 *q = side + lane;
    V4 acc = {0, 0, 0, 0};
    size_t sr = (size_t) wave * (side_rows / 4);
    for (size_t r = 0; r < rounds; ++r) {
        V4 v[10], w[SIDE > 0 ? SIDE : 1];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = __builtin_nontemporal_load(p + (r * 10 + u) * 64);
#pragma unroll
        for (int u = 0; u < SIDE; ++u) { w[u] = q[sr * 64]; sr = sr + 1 < side_rows ? sr + 1 : 0; }
#pragma unroll
        for (int u = 0; u < 10; ++u) acc += v[u];
#pragma unroll
        for (int u = 0; u < SIDE; ++u) acc += w[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <int SIDE>
static void run(const V4 *buf, size_t bytes, const V4 *side, float *out) {
    const size_t rounds = bytes / (1000 * 10 * 1024);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    for (int f = 0; f < 5; ++f) hipLaunchKernelGGL((k<SIDE>), dim3(250), dim3(256), 0, 0, buf, rounds, side, (size_t) 1250, out);
    (void) hipEventRecord(e0, 0);
    for (int f = 0; f < 40; ++f) hipLaunchKernelGGL((k<SIDE>), dim3(250), dim3(256), 0, 0, buf, rounds, side, (size_t) 1250, out);
    (void) hipEventRecord(e1, 0); (void) hipEventSynchronize(e1);
    float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 40, b = (double) rounds * 1000 * 10 * 1024;
    printf("side %d KB per 10 KB of stream: %7.1f us per frame, stream %5.2f TB/s, stream + side %5.2f TB/s\n", SIDE, us, b / us / 1e6, b * (10 + SIDE) / 10 / us / 1e6);
}
int main() {
    V4 *buf, *side; float *out;
    const size_t bytes = (size_t) 800 << 20;
    (void) hipMalloc(&buf, bytes); (void) hipMalloc(&side, 2 << 20); (void) hipMalloc(&out, 64);
    (void) hipMemset(buf, 0, bytes); (void) hipMemset(side, 0, 2 << 20);
    run<0>(buf, bytes, side, out); run<2>(buf, bytes, side, out); run<4>(buf, bytes, side, out); run<8>(buf, bytes, side, out);
    return 0;
}
