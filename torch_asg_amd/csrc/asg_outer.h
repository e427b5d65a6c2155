// torch_asg_amd/csrc/asg_outer.h -- sum over frames of outer products u_t (x) v_t on the matrix cores, exact fp32.
//
// The transition gradient of the full lattice is  sum_t  u_t[i] * v_t[j]  (u = posterior / row sum, v = the vector that
// went into the recursion's mat-vec; scaled by E[i][j] once at the end: DESIGN.md section 3).  That IS a dense
// contraction over the frame axis, K = number of frames, so it goes to v_mfma_f32_16x16x4_f32: exact fp32 (a k-ordered
// fmaf chain, bit-identical run to run), 4 frames per instruction, on a pipe the recursion wavefronts do not use.
// The recursion itself (a mat-vec per dependent step) stays on the VALU.
//
// Layout: a wavefront holds one frame's vector with element i in lane i (i < 64).  Four frames' vectors X0..X3 are
// turned into MFMA operands with two rounds of half/row swaps (a 4x4 transpose of 16-lane rows):
//   Y_r = [X0.row r | X1.row r | X2.row r | X3.row r]        (row = 16 lanes)
// which is exactly the A operand of tile-row r (lane l: A[i = l & 15][k = l >> 4]) and the B operand of tile-column r.
// Accumulator tile (r, c), register q, lane l holds  acc[16 r + 4 (l >> 4) + q][16 c + (l & 15)].
#pragma once
#include "asg_common.h"

namespace asg {

// v_permlane32_swap: lanes 32-63 of `a` swap with lanes 0-31 of `b`.
__device__ __forceinline__ void swap_halves(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// v_permlane16_swap: rows 1 and 3 (lanes 16-31, 48-63) of `a` swap with rows 0 and 2 (lanes 0-15, 32-47) of `b`.
__device__ __forceinline__ void swap_rows(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// In: x[k] = vector of frame k (element i in lane i).  Out: x[r] = operand of 16-row block r (see header).
__device__ __forceinline__ void frames_to_operands(float (&x)[4]) {
    swap_halves(x[0], x[2]);      // x0 = [a0 a1 c0 c1]   x2 = [a2 a3 c2 c3]      (a..d = frames 0..3, digit = row)
    swap_halves(x[1], x[3]);      // x1 = [b0 b1 d0 d1]   x3 = [b2 b3 d2 d3]
    swap_rows(x[0], x[1]);        // x0 = [a0 b0 c0 d0]   x1 = [a1 b1 c1 d1]
    swap_rows(x[2], x[3]);        // x2 = [a2 b2 c2 d2]   x3 = [a3 b3 c3 d3]
}

// acc[r][c] += sum_{k<4} u_k[16r..16r+15] (x) v_k[16c..16c+15];  NT = ceil(N / 16) tiles per side.
template <int NT>
__device__ __forceinline__ void outer4_accumulate(float (&u)[4], float (&v)[4], V4<float> (&acc)[NT * NT]) {
    frames_to_operands(u);
    frames_to_operands(v);
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int c = 0; c < NT; ++c)
            acc[r * NT + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[r], v[c], acc[r * NT + c], 0, 0, 0);
}

// The same sum with double accumulators (v_mfma_f64_16x16x4_f64, operands widened after the transposes): a hundred
// products of ~1 summed in fp32 lose 4e-6 per add, which tiny alphabets -- whose transition gradient is an exact
// cancellation against the aligned lattice -- expose.  Accumulator tile (r, c), register q, lane l holds
// acc[16 r + (l >> 4) + 4 q][16 c + (l & 15)]   (NOT the fp32 form's row mapping).
typedef double V4d __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void outer4_accumulate_f64(float (&u)[4], float (&v)[4], V4d (&acc)[NT * NT]) {
    frames_to_operands(u);
    frames_to_operands(v);
    double ud[NT], vd[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) { ud[r] = (double) u[r]; vd[r] = (double) v[r]; }
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int c = 0; c < NT; ++c)
            acc[r * NT + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ud[r], vd[c], acc[r * NT + c], 0, 0, 0);
}

}  // namespace asg
