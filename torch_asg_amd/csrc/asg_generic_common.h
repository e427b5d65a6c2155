// torch_asg_amd/csrc/asg_generic_common.h -- what the four translation units of the generic path (large alphabets N > 64 and / or long
// targets S > 64) share: small device helpers, the split of a float into three bfloat16, sizes and offsets of the forward work area, and
// the declarations of the per-unit launchers.  Round 6 split asg_generic.hip (4 000 lines, one translation unit, 70 s of hipcc) into
//   asg_generic_step.hip     full lattice, forward: transition prep, per-frame step kernels (VALU / 16 x 16 tiles / fp32 MFMA stream),
//                            medium-alphabet kernel, resident-slice kernel + its in-stream repair, scores; sizes of the work area
//   asg_generic_aligned.hip  force-aligned lattice with long targets: forward kernels (long / pipe / wide / strip) and their gradient
//   asg_generic_grad.hip     full lattice, gradient: posterior rows, the contraction over the frame axis (fp32 MFMA, three-bfloat16
//                            planes on the bf16 pipe), exact fix-ups; backward scratch layout
//   asg_generic.hip          launch_fwd_generic / launch_bwd_generic: which of the above a problem takes
#pragma once
#include <cstdio>
#include <cstdlib>
#include "asg_common.h"
#include "asg_outer.h"
#include "asg_kernels.h"

namespace asg {

namespace {

__device__ __forceinline__ int gclampi(int64_t v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int) v); }

// order-preserving float <-> uint key (for atomicMax of a float-valued normaliser)
__device__ __forceinline__ unsigned fkey(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float funkey(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

template <typename R>
__device__ __forceinline__ R block_reduce_max(R v, R *red) {   // 256 threads; red[4]
    v = wave_allmax(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    R r = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    return r;
}
template <typename R>
__device__ __forceinline__ R block_reduce_sum(R v, R *red) {
    v = wave_allsum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    R r = (red[0] + red[1]) + (red[2] + red[3]);
    return r;
}

inline size_t au(size_t x) { return (x + 255) & ~(size_t) 255; }
typedef float V4f __attribute__((ext_vector_type(4)));

// A float as the exact sum of three bfloat16 (8 significant bits each, round to nearest at every step: the remainders are exact
// in fp32): two floats -> three packed bfloat16 pairs with v_cvt_pk_bf16_f32, the pair widened again (shift / mask), one packed
// subtraction per level.  Used by the large-alphabet gradient contraction (gemm3_pack_kernel).
typedef __bf16 BF8 __attribute__((ext_vector_type(8)));
typedef __bf16 BF2 __attribute__((ext_vector_type(2)));
typedef float F2v __attribute__((ext_vector_type(2)));
typedef unsigned U4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3x2(float x, float y, unsigned &h, unsigned &m, unsigned &l) {
    const F2v v = {x, y};
    const BF2 bh = __builtin_convertvector(v, BF2);
    const F2v r1 = v - __builtin_convertvector(bh, F2v);
    const BF2 bm = __builtin_convertvector(r1, BF2);
    const F2v r2 = r1 - __builtin_convertvector(bm, F2v);
    const BF2 bl = __builtin_convertvector(r2, BF2);
    h = __builtin_bit_cast(unsigned, bh); m = __builtin_bit_cast(unsigned, bm); l = __builtin_bit_cast(unsigned, bl);
}

// fp32 takes the matrix-core step (fwd_step_mfma) and its operand-order copies; fp64 the VALU / tile kernels
template <typename R> struct StepUsesMfma { static constexpr bool v = false; };
#ifndef ASG_X_NO_STEP_MFMA
template <> struct StepUsesMfma<float> { static constexpr bool v = true; };
#endif

// offset of the alpha pass's per-frame normaliser log inside the forward work area (its last member; asg_generic_step.hip lays the area out)
inline size_t work_mulog_offset(size_t elem, int T, int B, int npad) {
    return au((size_t) T * B * elem) + 2 * au(2 * (size_t) B * npad * elem) + 2 * au(3 * (size_t) B * 4) + 2 * au((size_t) B * 8);
}

// frames per workgroup of the gradient kernels that walk (utterance, chunk of frames)
inline void generic_chunks(int T, int B, int *chunk, int *nchunks) {
    int nch = (512 + B - 1) / B;
    if (nch < 1) nch = 1;
    int ch = (T + nch - 1) / nch;
    if (ch < 16) ch = 16;
    ch = (ch + 3) / 4 * 4;
    *chunk = ch;
    *nchunks = (T + ch - 1) / ch;
}

}  // namespace

// ---- per-unit launchers (asg_generic.hip dispatches) --------------------------------------------------------------------
template <typename R>
hipError_t launch_fwd_full_generic(const Problem &P, const State &W, const FwdOut &O, int full_mask, bool store, hipStream_t stream);
template <typename R>
hipError_t launch_fwd_aligned_generic(const Problem &P, const State &W, const FwdOut &O, int ali_mask, bool store, hipStream_t stream);

// where the pieces of the backward scratch buffer are (bwd_scratch_bytes_generic sizes it; asg_generic_grad.hip)
struct GenericBwdLayout {
    void *Pm, *Gm;              // [rows][npad] posterior rows / exp-domain previous frame
    void *gHD;                  // [B][nchunks][2][S] aligned edge posteriors
    int *anybad, *rowoff;
    void *atiles;               // per-chunk tiles (N <= 64) or the 64-bit fixed-point accumulator (N <= 2048)
    void *gpart;                // partial sums of a sliced contraction
    unsigned short *planes3;    // bfloat16 planes of the contraction's operands
    int npad;
};
GenericBwdLayout generic_bwd_layout(size_t elem, const Problem &P, const BwdArgs &A);
// *fx_cleared: the fixed-point accumulator of the aligned part was cleared together with the flag word
template <typename R>
hipError_t launch_bwd_full_generic(const Problem &P, const State &W, const BwdArgs &A, const GenericBwdLayout &Y, bool do_ali, bool *fx_cleared,
                                   hipStream_t stream);
template <typename R>
hipError_t launch_bwd_aligned_generic(const Problem &P, const State &W, const BwdArgs &A, const GenericBwdLayout &Y, bool have_full,
                                      bool fx_cleared, hipStream_t stream);

}  // namespace asg
