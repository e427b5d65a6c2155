// Developer micro-benchmark (gfx950), round 6 / VERDICT r5 item 6: the PRODUCT LOOP of a one-copy large-alphabet step, alone.
//
// The shipped step streams two normalised copies of the transition matrix per frame (800 MB at N = 10^4: 130 us at 6.2 TB/s).  The
// one-copy form would keep ONE copy as three bfloat16 planes (600 MB: every float the exact sum of three bfloat16, six partial products
// per product on v_mfma_f32_32x32x16_bf16) in 2-D tiles, one 640 x 640 tile per compute unit, walked in 64 x 64 panels; every panel is
// used twice from LDS: alpha[i] += G[i][j] v[j] with the fragment as stored (ds_read_b128), beta[j] += G[i][j] y[i] with the TRANSPOSED
// fragment (ds_read_b64_tr_b16).  This file times exactly that loop -- no epilogue, no exchange of the partial sums between tiles:
//
//   mode 0   the 24 matrix instructions of a 32 x 32 block with operands that never change (the pipe's rate)
//   mode 1   + the block's fragment reads from a panel that sits in LDS (6 ds_read_b128 + 12 ds_read_b64_tr_b16 + 6 ds_read_b128 for v)
//   mode 2   + one workgroup barrier per panel (what double-buffered panels need)
//   mode 3   + the panel STREAM: every panel's 24 KB arrive by LDS-DMA (buffer_load_dwordx4 .. lds) from a buffer larger than the
//            memory-side cache, double-buffered; 100 panels per workgroup and "frame" = the 600 MB of cfg 5's matrix over 256 compute units
//   mode 4   the stream alone (mode 3 without fragment reads and products)
// each with one workgroup (4 wavefronts: one per SIMD) or two workgroups per compute unit.
//
// Fragment addresses follow the layout the real kernel would use ([plane][group of 8 columns][row][8]: a wavefront's ds_read_b128 is
// lane-linear; the transposed reads take 8 bytes of 64 distinct 16-byte units) -- the VALUES are not checked: this is a rate probe.
// build: hipcc -O3 --offload-arch=gfx950 onecopy_loop.hip -o onecopy_loop ; run: ./onecopy_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __bf16 BF8 __attribute__((ext_vector_type(8)));
typedef __bf16 BF4 __attribute__((ext_vector_type(4)));
typedef float V16f __attribute__((ext_vector_type(16)));
typedef unsigned U4 __attribute__((ext_vector_type(4)));

constexpr int kPlaneUnits = 8 * 64;                  // 16-byte units of one plane of a 64 x 64 panel: 8 column groups x 64 rows
constexpr int kPanelUnits = 3 * kPlaneUnits;         // 24 KB
constexpr int kVecUnits = 3 * 8 * 32;                // v of a panel's 64 columns: 3 planes x 8 column groups x 32 utterances (12 KB)
constexpr size_t kLdsBytes = (size_t) (2 * kPanelUnits + kVecUnits) * 16;

// JB = 32-column blocks a wavefront owns beta accumulators for: 10 (a 640-wide tile, one workgroup per compute unit: 276 registers) or 5 (two
// workgroups per compute unit, a 320-wide tile each: 196 registers, two wavefronts per SIMD)
template <int MODE, int kJB>
__global__ void __launch_bounds__(256) k(const U4 *src, size_t units_per_wg_frame, size_t wrap_units, float *out, long long *clk, int panels) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    U4 *lds = reinterpret_cast<U4 *>(raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int ib = wave & 1, jb = wave >> 1, m = lane & 31, kg = lane >> 5;
    // something finite in every unit
    for (int u = threadIdx.x; u < 2 * kPanelUnits + kVecUnits; u += 256) lds[u] = U4{0x3c003c00u + (unsigned) u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    V16f accA = {}, accB[kJB];
#pragma unroll
    for (int q = 0; q < kJB; ++q) accB[q] = V16f{};
    BF8 yf[2][3];                                     // y of this wavefront's 32 rows: stays in registers along a row of panels
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) yf[ks][pl] = __builtin_bit_cast(BF8, lds[2 * kPanelUnits + pl * 256 + (2 * ks + kg) * 32 + m]);
    // the stream: workgroup b reads its own contiguous region, 24 transfers of 1 KB per panel, 6 per wavefront
    const size_t wg_base = (size_t) blockIdx.x * units_per_wg_frame;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) src, 0, 0xffffffffu, 0x00020000);
    // panel n of this workgroup: frame n / ppf (three frames' worth of buffer, cycled), the workgroup's own contiguous slice of it
    const int ppf = (int) (units_per_wg_frame / kPanelUnits);
    const size_t frame_units = wrap_units / 3;
    int pidx = 0;
    size_t cursor = wg_base;
    auto dma = [&](int stage, size_t at_unit) {
        const unsigned base = (unsigned) (uintptr_t) (__attribute__((address_space(3))) void *) (lds + stage * kPanelUnits) + 1024u * 6u * (unsigned) wave;
        const unsigned v = (unsigned) lane * 16u;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const unsigned l = __builtin_amdgcn_readfirstlane(base + 1024u * d);
            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned) ((at_unit + (size_t) (6 * wave + d) * 64) * 16));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(v), "s"(rs), "s"(l), "s"(so) : "memory", "m0");
        }
    };
    auto next_cursor = [&]() {
        ++pidx;          // (629 MB go by between two visits of a frame's slice: far beyond the 256 MiB memory-side cache)
        cursor = (size_t) ((pidx / ppf) % 3) * frame_units + wg_base + (size_t) (pidx % ppf) * kPanelUnits;
    };
    if (MODE >= 3) { dma(0, cursor); next_cursor(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    const long long t0 = clock64();
    for (int p0 = 0; p0 < panels; p0 += kJB) {
#pragma unroll
        for (int q = 0; q < kJB; ++q) {
            const int stage = (MODE >= 3) ? ((p0 + q) & 1) : 0;
            if (MODE >= 3) { dma(stage ^ 1, cursor); next_cursor(); }
            const U4 *cur = lds + stage * kPanelUnits;
            if (MODE != 4) {
                BF8 af[2][3], vf[2][3], gt[2][3];
                if (MODE >= 1) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            af[ks][pl] = __builtin_bit_cast(BF8, cur[pl * kPlaneUnits + (4 * jb + 2 * ks + kg) * 64 + 32 * ib + m]);
                            vf[ks][pl] = __builtin_bit_cast(BF8, lds[2 * kPanelUnits + pl * 256 + (4 * jb + 2 * ks + kg) * 32 + m]);
                            // transposed fragment: two 8-byte reads, each lane its own 16-byte unit (column group by lane & 3, row by the rest)
                            const unsigned char *b0 = reinterpret_cast<const unsigned char *>(
                                cur + pl * kPlaneUnits + (4 * jb + (lane & 3)) * 64 + 32 * ib + 16 * ks + ((lane >> 2) & 7) + 8 * (lane >> 5));
                            const BF4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) BF4 *) (b0));
                            const BF4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) BF4 *) (b0 + 8));
                            gt[ks][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                } else {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) { af[ks][pl] = yf[ks][pl]; vf[ks][pl] = yf[ks ^ 1][pl]; gt[ks][pl] = yf[ks][(pl + 1) % 3]; }
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // alpha: the stored fragment times v (smallest terms first), 6 products
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][2], vf[ks][0], accA, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], vf[ks][2], accA, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], vf[ks][1], accA, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][1], vf[ks][0], accA, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], vf[ks][1], accA, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][0], vf[ks][0], accA, 0, 0, 0);
                    // beta: the transposed fragment times y
                    accB[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[ks][2], yf[ks][0], accB[q], 0, 0, 0);
                    accB[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[ks][0], yf[ks][2], accB[q], 0, 0, 0);
                    accB[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[ks][1], yf[ks][1], accB[q], 0, 0, 0);
                    accB[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[ks][1], yf[ks][0], accB[q], 0, 0, 0);
                    accB[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[ks][0], yf[ks][1], accB[q], 0, 0, 0);
                    accB[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[ks][0], yf[ks][0], accB[q], 0, 0, 0);
                }
            }
            if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE >= 2) __syncthreads();
        }
    }
    const long long t1 = clock64();
    float s = accA[0] + accA[7];
#pragma unroll
    for (int q = 0; q < kJB; ++q) s += accB[q][0] + accB[q][15];
    out[(size_t) blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, int kJB>
void run(const char *what, const U4 *src, size_t src_units, int wgs_per_cu) {
    float *out; long long *clk;
    const int cus = 256, grid = cus * wgs_per_cu;
    (void) hipMalloc(&out, (size_t) grid * 256 * 4); (void) hipMalloc(&clk, 8);
    (void) hipFuncSetAttribute((const void *) k<MODE, kJB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kLdsBytes);
    // one "frame" = 100 panels per compute unit (16 x 16 tiles of 640 x 640 over 256 compute units: cfg 5's matrix once)
    const int frames = 20, panels = 100 * frames / wgs_per_cu;
    const size_t per_wg = (size_t) (100 / wgs_per_cu) * kPanelUnits;
    hipLaunchKernelGGL((k<MODE, kJB>), dim3(grid), dim3(256), kLdsBytes, 0, src, per_wg, src_units, out, clk, 20);
    (void) hipDeviceSynchronize();
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, kJB>), dim3(grid), dim3(256), kLdsBytes, 0, src, per_wg, src_units, out, clk, panels);
    (void) hipEventRecord(e1, 0);
    (void) hipDeviceSynchronize();
    float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1);
    long long c = 0; (void) hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double us_frame = ms * 1e3 / frames;
    const double mfma_per_simd_frame = 100.0 * 24.0;           // 100 panels x one 32 x 32 block per SIMD x 24 products
    printf("%-74s %d wg/cu: %7.1f us per frame", what, wgs_per_cu, us_frame);
    if (MODE != 4) printf("  %5.1f cycles per matrix instruction at 2.4 GHz (wave 0 by its own clock: %5.1f ticks)", us_frame * 2400.0 / mfma_per_simd_frame,
                          (double) c / ((double) panels * 24.0));
    if (MODE >= 3) printf("  stream %.2f TB/s", 256.0 * 100.0 * kPanelUnits * 16.0 / (us_frame * 1e-6) / 1e12);
    printf("\n");
    (void) hipFree(out); (void) hipFree(clk);
}

int main() {
    hipDeviceProp_t prop; (void) hipGetDeviceProperties(&prop, 0);
    printf("%s, %d compute units, clock %d MHz; one frame = 100 panels of 64 x 64 x 3 bfloat16 planes per compute unit (629 MB over 256)\n",
           prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const size_t src_units = (size_t) 3 * 256 * 100 * kPanelUnits;           // three frames: 1.9 GB
    U4 *src; (void) hipMalloc(&src, src_units * 16);
    (void) hipMemset(src, 0x3c, src_units * 16);
#define ALL(JB_, W_) \
    run<0, JB_>("0 products only, operands in registers (24 per 32 x 32 block)", src, src_units, W_); \
    run<1, JB_>("1 + fragment reads from a resident panel (12 b128 + 12 tr_b16 per block)", src, src_units, W_); \
    run<2, JB_>("2 + a workgroup barrier per panel", src, src_units, W_); \
    run<3, JB_>("3 + the panel stream by LDS-DMA, double-buffered (the loop of the kernel)", src, src_units, W_); \
    run<4, JB_>("4 the stream alone (LDS-DMA + barrier per panel, no reads, no products)", src, src_units, W_);
    ALL(10, 1)
    ALL(5, 2)
    printf("reference points: the bf16 pipe issues a 32x32x16 product every 32 cycles per SIMD (30.5 us per frame for 2400 of them);\n"
           "the shipped two-copy fp32 step takes 130 us per frame (805 MB at 6.2 TB/s); 600 MB at that rate are 97 us.\n");
    return 0;
}
