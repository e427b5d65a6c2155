// Developer micro-benchmark (gfx950), second version of blocked_matvec.hip: the same 8 x 8 lane-grid mat-vec with
//   * packed FMAs (two outputs per v_pk_fma_f32, the input broadcast through op_sel),
//   * the cross-row all-reduce of step B as reduce-scatter (swaps) -> row_ror:8 -> all-gather (swaps),
//   * the realistic extras of the recursion wavefront, branch-free: emission factors read from an LDS ring one step
//     ahead (b128 + b32 per step), row sums exported to an LDS ring by 8 lanes (the others write to a dump row),
//     16 steps unrolled so that every ring offset is an instruction immediate.
// build: hipcc -O3 --offload-arch=gfx950 blocked_matvec2.hip -o blocked_matvec2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#ifndef NP
#define NP 40
#endif
constexpr int BS = NP / 8;
constexpr int BP = (BS + 1) / 2;          // output pairs
typedef float V2 __attribute__((ext_vector_type(2)));
typedef float V4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void swap32(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kXor1 = 0xB1, kXor2 = 0x4E, kHalfMirror = 0x141, kRor8 = 0x128;

// acc[p] = (out 2p, out 2p+1) = sum_b E2[p][b] * x[b]
__device__ __forceinline__ void partial(const V2 (&E2)[BP][BS], const float (&x)[BS], V2 (&acc)[BP]) {
#pragma unroll
    for (int p = 0; p < BP; ++p) acc[p] = E2[p][0] * V2{x[0], x[0]};
#pragma unroll
    for (int b = 1; b < BS; ++b)
#pragma unroll
        for (int p = 0; p < BP; ++p) acc[p] = __builtin_elementwise_fma(E2[p][b], V2{x[b], x[b]}, acc[p]);
}
__device__ __forceinline__ void unpack(const V2 (&acc)[BP], float (&v)[2 * BP]) {
#pragma unroll
    for (int p = 0; p < BP; ++p) { v[2 * p] = acc[p].x; v[2 * p + 1] = acc[p].y; }
}
template <int K>
__device__ __forceinline__ void reduce_c(float (&v)[K]) {      // all-reduce over lane bits 0..2
#pragma unroll
    for (int a = 0; a < BS; ++a) v[a] += dpp<kXor1>(v[a]);
#pragma unroll
    for (int a = 0; a < BS; ++a) v[a] += dpp<kXor2>(v[a]);
#pragma unroll
    for (int a = 0; a < BS; ++a) v[a] += dpp<kHalfMirror>(v[a]);
}
// all-reduce over lane bits 3..5 of BS (4..6) values: values 0..3 through a reduce-scatter over bits 5, 4 (3 swaps, 3 adds),
// the bit-3 stage on the ONE scattered register, then an all-gather (3 copies, 3 swaps); values 4, 5 as a pair
template <int K>
__device__ __forceinline__ void reduce_r(float (&v)[K]) {
    swap32(v[0], v[1]); float z01 = v[0] + v[1];
    swap32(v[2], v[3]); float z23 = v[2] + v[3];
    swap16(z01, z23); float w = z01 + z23;          // rows: X0 X2 X1 X3
    w += dpp<kRor8>(w);
    float t = w; swap16(w, t);                      // w = [X0 X0 X1 X1], t = [X2 X2 X3 X3]
    float w2 = w, t2 = t;
    swap32(w, w2);
    swap32(t, t2);
    v[0] = w; v[1] = w2; v[2] = t; v[3] = t2;
    if constexpr (BS == 5) {
        float x = v[4] + dpp<kRor8>(v[4]);
        float y = x; swap16(x, y); x += y;
        y = x; swap32(x, y); v[4] = x + y;
    } else if constexpr (BS == 6) {
        swap32(v[4], v[5]); float z = v[4] + v[5];  // halves: X4 | X5
        z += dpp<kRor8>(z);
        float y = z; swap16(z, y); z += y;
        y = z; swap32(z, y);                        // z = X4 everywhere, y = X5 everywhere
        v[4] = z; v[5] = y;
    }
}

__device__ __forceinline__ void mul_e(const float (&v)[2 * BP], const float (&ef)[8], float (&x)[BS]) {
#pragma unroll
    for (int p = 0; p < BS / 2; ++p) {
        const V2 q = V2{v[2 * p], v[2 * p + 1]} * V2{ef[2 * p], ef[2 * p + 1]};
        x[2 * p] = q.x; x[2 * p + 1] = q.y;
    }
    if (BS & 1) x[BS - 1] = v[BS - 1] * ef[BS - 1];
}

template <bool EXTRA>
__global__ void __launch_bounds__(64, 1) k(const float *E, const float *v0, float *out, long long *clk, int iters, int N) {
    __shared__ __attribute__((aligned(16))) float ering[16][64];     // [slot][8 groups][8]
    __shared__ __attribute__((aligned(16))) float sring[32][64];     // rows 16..31: dump
    const int lane = threadIdx.x;
    const int r = lane >> 3, c = lane & 7;
    for (int q = lane; q < 16 * 64; q += 64) (&ering[0][0])[q] = 0.5f;
    for (int q = lane; q < 32 * 64; q += 64) (&sring[0][0])[q] = 0.f;
    __syncthreads();
    V2 EA[BP][BS], EB[BP][BS];
    float x[BS];
    auto el = [&](int i, int j) { return (i < N && j < N) ? E[i * N + j] : 0.f; };
    for (int p = 0; p < BP; ++p)
        for (int b = 0; b < BS; ++b) {
            const int a0 = 2 * p, a1 = 2 * p + 1;
            EA[p][b] = V2{el(BS * r + a0, BS * c + b), a1 < BS ? el(BS * r + a1, BS * c + b) : 0.f};
            EB[p][b] = V2{el(BS * c + a0, BS * r + b), a1 < BS ? el(BS * c + a1, BS * r + b) : 0.f};
        }
    for (int b = 0; b < BS; ++b) x[b] = (BS * c + b < N) ? v0[BS * c + b] : 0.f;
    // LDS byte offsets: emission factors of group r (step A) / c (step B); export rows: real for 8 lanes, dump for the rest
    const float *epA = &ering[0][8 * r], *epB = &ering[0][8 * c];
    float *spA = (c == 0) ? &sring[0][8 * r] : &sring[16][0], *spB = (r == 0) ? &sring[0][8 * c] : &sring[16][0];
    long long t0 = clock64();
    for (int it = 0; it < iters; it += 16) {
        float ef[8];
        if (EXTRA) {
            const V4 q = *reinterpret_cast<const V4 *>(epA);
            ef[0] = q.x; ef[1] = q.y; ef[2] = q.z; ef[3] = q.w; ef[4] = epA[4];
        }
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            // ---- step A (slot j)
            {
                float en[8];
                if (EXTRA) {
                    const V4 q = *reinterpret_cast<const V4 *>(epB + (j + 1) * 64);
                    en[0] = q.x; en[1] = q.y; en[2] = q.z; en[3] = q.w; en[4] = epB[(j + 1) * 64 + 4];
                } else {
#pragma unroll
                    for (int a = 0; a < BS; ++a) { ef[a] = 0.5f; en[a] = 0.5f; }
                }
                V2 acc[BP];
                partial(EA, x, acc);
                float v[2 * BP];
                unpack(acc, v);
                reduce_c(v);
                if (EXTRA) {
                    float *sp = spA + j * 64;
                    *reinterpret_cast<V4 *>(sp) = V4{v[0], v[1], v[2], v[3]};
                    sp[4] = v[4];
                }
                mul_e(v, ef, x);
#pragma unroll
                for (int a = 0; a < BS; ++a) ef[a] = en[a];
            }
            // ---- step B (slot j + 1)
            {
                float en[8];
                if (EXTRA) {
                    const int jn = (j + 2) & 15;
                    const V4 q = *reinterpret_cast<const V4 *>(epA + jn * 64);
                    en[0] = q.x; en[1] = q.y; en[2] = q.z; en[3] = q.w; en[4] = epA[jn * 64 + 4];
                } else {
#pragma unroll
                    for (int a = 0; a < BS; ++a) en[a] = 0.5f;
                }
                V2 acc[BP];
                partial(EB, x, acc);
                float v[2 * BP];
                unpack(acc, v);
                reduce_r(v);
                if (EXTRA) {
                    float *sp = spB + (j + 1) * 64;
                    *reinterpret_cast<V4 *>(sp) = V4{v[0], v[1], v[2], v[3]};
                    sp[4] = v[4];
                }
                mul_e(v, ef, x);
#pragma unroll
                for (int a = 0; a < BS; ++a) ef[a] = en[a];
            }
        }
    }
    long long t1 = clock64();
    if (r == 0) for (int b = 0; b < BS; ++b) out[BS * c + b] = x[b];
    if (lane == 0) clk[0] = t1 - t0;
    if (EXTRA && lane == 1) clk[1] = (long long) sring[3][5];
}

template <bool EXTRA>
void run(const float *dE, const float *dv, float *dout, long long *dclk, const std::vector<float> &E, const std::vector<float> &v, int N) {
    for (int iters : {16, 4000}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<EXTRA>), dim3(1), dim3(64), 0, 0, dE, dv, dout, dclk, iters, N);
        (void) hipDeviceSynchronize();
        float out[64]; long long clk;
        (void) hipMemcpy(out, dout, 64 * 4, hipMemcpyDeviceToHost);
        (void) hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
        if (iters == 16) {
            std::vector<double> p(v.begin(), v.end()), s(N);
            for (int it = 0; it < iters; ++it) {
                for (int i = 0; i < N; ++i) { double a = 0; for (int j = 0; j < N; ++j) a += (double) E[i * N + j] * p[j]; s[i] = a; }
                for (int i = 0; i < N; ++i) p[i] = s[i] * 0.5;
            }
            double worst = 0;
            for (int i = 0; i < N; ++i) worst = fmax(worst, fabs(out[i] - p[i]) / fabs(p[i]));
            printf("NP %d extra %d: worst relative error after 16 steps %.2e\n", NP, (int) EXTRA, worst);
        } else {
            printf("NP %d extra %d: %.1f cycles/step\n", NP, (int) EXTRA, (double) clk / iters);
        }
    }
}

int main() {
    const int N = NP;
    std::vector<float> E(N * N), v(N);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) E[i * N + j] = 0.02f + 0.03f * ((i * 7 + j * 13) % 11) / 11.f;
        v[i] = 1.0f + 0.01f * i;
    }
    float *dE, *dv, *dout; long long *dclk;
    (void) hipMalloc(&dE, N * N * 4); (void) hipMalloc(&dv, N * 4); (void) hipMalloc(&dout, 64 * 4); (void) hipMalloc(&dclk, 16);
    (void) hipMemcpy(dE, E.data(), N * N * 4, hipMemcpyHostToDevice);
    (void) hipMemcpy(dv, v.data(), N * 4, hipMemcpyHostToDevice);
    run<false>(dE, dv, dout, dclk, E, v, N);
    run<true>(dE, dv, dout, dclk, E, v, N);
    return 0;
}
