// torch_asg_amd/csrc/asg_common.h -- device helpers shared by the gfx950 ASG kernels.
//
// Conventions used by every kernel in this directory:
//  * all lattice state is kept in LOG2 units (x2 = x * log2(e)) so the per-node
//    transcendental is a bare v_exp_f32 / v_log_f32;
//  * a wavefront is 64 lanes; cross-lane traffic uses DPP / v_readlane, never ds_bpermute;
//  * per-frame state is stored RELATIVE to a running offset (kept in double), which is
//    what gives better-than-reference fp32 accuracy for long utterances.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asg {

constexpr int kWave = 64;

template <typename R> struct Num;

template <> struct Num<float> {
    static __device__ __forceinline__ float ninf() { return -__builtin_inff(); }
    static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
    static __device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
    static __device__ __forceinline__ float log2e() { return 1.4426950408889634f; }
    static __device__ __forceinline__ float tiny() { return 1e-30f; }     // below this the exp-domain sum is re-done exactly
    static __device__ __forceinline__ float ls_floor() { return -100.0f; }
    static constexpr double kFix = 1099511627776.0;                       // 2^40 fixed-point scale
};

template <> struct Num<double> {
    static __device__ __forceinline__ double ninf() { return -__builtin_inf(); }
    static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
    static __device__ __forceinline__ double log2e() { return 1.4426950408889634; }
    static __device__ __forceinline__ double tiny() { return 1e-280; }
    static __device__ __forceinline__ double ls_floor() { return -900.0; }
    static constexpr double kFix = 17592186044416.0;                      // 2^44
};

constexpr double kLn2 = 0.6931471805599453;

// ---- DPP moves -----------------------------------------------------------
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl=false): lanes whose source
// is out of range keep `old`.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float oldv, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double oldv, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(oldv), __double2loint(v), CTRL, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(oldv), __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

constexpr int kDppXor1 = 0xB1;          // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // row_half_mirror
constexpr int kDppMirror = 0x140;       // row_mirror
constexpr int kDppWaveShl1 = 0x130;     // lane i <- lane i+1
constexpr int kDppWaveShr1 = 0x138;     // lane i <- lane i-1

__device__ __forceinline__ float readlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double readlane(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// value of lane i-1 (lane 0 gets `fill`) / lane i+1 (lane 63 gets `fill`)
template <typename R> __device__ __forceinline__ R from_prev_lane(R v, R fill) { return dpp_mov<kDppWaveShr1>(fill, v); }
template <typename R> __device__ __forceinline__ R from_next_lane(R v, R fill) { return dpp_mov<kDppWaveShl1>(fill, v); }

// All-lanes max / sum of one value per lane over the first ROWS*16 lanes (lanes beyond
// must hold the identity).  4 DPP steps inside each 16-lane row, then v_readlane per row.
template <int ROWS, typename R>
__device__ __forceinline__ R wave_allmax(R v) {
    v = fmax(v, dpp_mov<kDppXor1>(v, v));
    v = fmax(v, dpp_mov<kDppXor2>(v, v));
    v = fmax(v, dpp_mov<kDppHalfMirror>(v, v));
    v = fmax(v, dpp_mov<kDppMirror>(v, v));
    R r = readlane(v, 0);
    if (ROWS > 1) r = fmax(r, readlane(v, 16));
    if (ROWS > 2) r = fmax(r, readlane(v, 32));
    if (ROWS > 3) r = fmax(r, readlane(v, 48));
    return r;
}
template <int ROWS, typename R>
__device__ __forceinline__ R wave_allsum(R v) {
    v += dpp_mov<kDppXor1>(v, v);
    v += dpp_mov<kDppXor2>(v, v);
    v += dpp_mov<kDppHalfMirror>(v, v);
    v += dpp_mov<kDppMirror>(v, v);
    R r = readlane(v, 0);
    if (ROWS > 1) r += readlane(v, 16);
    if (ROWS > 2) r += readlane(v, 32);
    if (ROWS > 3) r += readlane(v, 48);
    return r;
}

// log2(2^a + 2^b); (-inf,-inf) -> -inf
template <typename R>
__device__ __forceinline__ R lse2(R a, R b) {
    R m = fmax(a, b);
    R d = fmin(a, b) - m;                       // <= 0, NaN when both are -inf
    R r = m + Num<R>::log2(R(1) + Num<R>::exp2(d));
    return (m == Num<R>::ninf()) ? m : r;
}

// fixed-point helpers for deterministic LDS scatter-adds (integer adds commute)
template <typename R> __device__ __forceinline__ unsigned long long to_fix(R x) {
    return (unsigned long long) __double2ll_rn((double) x * Num<R>::kFix);
}
template <typename R> __device__ __forceinline__ R from_fix(unsigned long long v) {
    return (R) ((double) (long long) v * (1.0 / Num<R>::kFix));
}

}  // namespace asg
