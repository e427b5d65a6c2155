// torch_asg_amd/csrc/asg_common.h -- device helpers shared by the gfx950 ASG kernels.
//
// Conventions used by every kernel in this directory:
//  * all lattice state is kept in LOG2 units (x2 = x * log2(e)) so the per-node
//    transcendental is a bare v_exp_f32 / v_log_f32;
//  * a wavefront is 64 lanes; cross-lane traffic uses DPP / v_readlane, never ds_bpermute;
//  * per-frame state is stored RELATIVE to a running offset (kept in double), which is
//    what gives better-than-reference fp32 accuracy for long utterances;
//  * "log zero" inside the recursions is any value <= kLogZero (-1e30): it absorbs finite addends,
//    never produces NaN in (a - b), and is mapped back to -inf at the API boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asg {

constexpr int kWave = 64;
constexpr double kLn2 = 0.6931471805599453;

// ---- zero-fill on a stream ---------------------------------------------------------------------------------
// Never hipMemsetAsync: a memset node recorded into a hipGraph writes its value at the FIRST replay only -- from the second replay on it
// writes its own node parameters (word count, a constant, a host pointer) instead -- on ROCm 7.2 / torch 2.10, any size, either API
// (tools/memset_in_graph_probe.py, profiles/r06_memset_in_graph.txt; found through a replayed stand-alone step that kept its first loss).
// A kernel node replays as recorded.  `p` 4-byte aligned, `bytes` a multiple of 4.
static __global__ void __launch_bounds__(256) zero_fill_kernel(unsigned *p, size_t nwords) {
    size_t head = ((16u - (unsigned) ((uintptr_t) p & 15u)) & 15u) / 4u;      // words up to the first 16-byte boundary
    if (head > nwords) head = nwords;
    const size_t tid = (size_t) blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t) gridDim.x * blockDim.x;
    if (tid < head) p[tid] = 0u;
    uint4 *q = reinterpret_cast<uint4 *>(p + head);
    const size_t n16 = (nwords - head) / 4;
    for (size_t i = tid; i < n16; i += nth) q[i] = make_uint4(0u, 0u, 0u, 0u);
    const size_t tail0 = head + 4 * n16;
    if (tid < nwords - tail0) p[tail0 + tid] = 0u;
}
static inline hipError_t zero_async(void *p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    const size_t nwords = (bytes + 3) / 4;
    size_t blocks = (nwords / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned) blocks), dim3(256), 0, stream, (unsigned *) p, nwords);
    return hipGetLastError();
}

template <typename R> using V2 = R __attribute__((ext_vector_type(2)));
template <typename R> using V4 = R __attribute__((ext_vector_type(4)));

template <typename R> struct Num;

template <> struct Num<float> {
    static __device__ __forceinline__ float ninf() { return -__builtin_inff(); }
    static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
    static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
    static __device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32
    static __device__ __forceinline__ float log2e() { return 1.4426950408889634f; }
    static __device__ __forceinline__ float logzero() { return -1e30f; }
    static __device__ __forceinline__ float lg_limit() { return 100.0f; }   // |log2(row sum)| beyond this -> exact path
    static constexpr double kFix = 1099511627776.0;                        // 2^40 fixed-point scale
};

template <> struct Num<double> {
    static __device__ __forceinline__ double ninf() { return -__builtin_inf(); }
    static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
    static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double log2e() { return 1.4426950408889634; }
    static __device__ __forceinline__ double logzero() { return -1e300; }
    static __device__ __forceinline__ double lg_limit() { return 900.0; }
    static constexpr double kFix = 17592186044416.0;                       // 2^44
};

// ---- DPP moves -----------------------------------------------------------
// BC = bound_ctrl: lanes whose source lane does not exist read 0 (BC) or keep `oldv` (!BC).
template <int CTRL, bool BC = false>
__device__ __forceinline__ float dpp_mov(float oldv, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(v), CTRL, 0xF, 0xF, BC));
}
template <int CTRL, bool BC = false>
__device__ __forceinline__ double dpp_mov(double oldv, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(oldv), __double2loint(v), CTRL, 0xF, 0xF, BC);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(oldv), __double2hiint(v), CTRL, 0xF, 0xF, BC);
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

// value of lane i-1 (lane 0 reads 0) / lane i+1 (lane 63 reads 0); fuses into the consuming VALU op
template <typename R> __device__ __forceinline__ R prev_lane_or_zero(R v) { return dpp_mov<kDppWaveShr1, true>(R(0), v); }
template <typename R> __device__ __forceinline__ R next_lane_or_zero(R v) { return dpp_mov<kDppWaveShl1, true>(R(0), v); }

// ---- wave-wide reductions, result uniform (SGPR) -------------------------
// fp32: six DPP-fused VALU ops (each needs 2 wait states after the VALU write it reads: s_nop 1;
// hipcc inserts nothing inside an asm block) + one v_readlane.
__device__ __forceinline__ float wave_allmax(float v) {
    float r;
    asm volatile(
        "s_nop 1\n"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "=&v"(r) : "v"(v));
    return readlane(r, 63);
}
__device__ __forceinline__ float wave_allsum(float v) {
    float r;
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "=&v"(r) : "v"(v));
    return readlane(r, 63);
}
// Two / three independent reductions interleaved in one block: the DPP wait states of one are filled by the
// other(s), so the latency is that of a single reduction.
#define ASG_DPP_STEP2(op, ctl) \
    op " %0, %0, %0 " ctl "\n" op " %1, %1, %1 " ctl "\n" "s_nop 0\n"
__device__ __forceinline__ void wave_allmax2(float &a, float &b) {
    float ra, rb;
    asm volatile(
        "s_nop 1\n"
        "v_max_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_max_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        ASG_DPP_STEP2("v_max_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP2("v_max_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP2("v_max_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP2("v_max_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        ASG_DPP_STEP2("v_max_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 0\n"
        : "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
    a = readlane(ra, 63);
    b = readlane(rb, 63);
}
__device__ __forceinline__ void wave_allsum2(float &a, float &b) {
    float ra, rb;
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        ASG_DPP_STEP2("v_add_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP2("v_add_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP2("v_add_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP2("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        ASG_DPP_STEP2("v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 0\n"
        : "=&v"(ra), "=&v"(rb) : "v"(a), "v"(b));
    a = readlane(ra, 63);
    b = readlane(rb, 63);
}

// Four independent fp32 sum reductions interleaved: with four chains in flight every DPP source is four instructions
// old, which satisfies the VALU-write -> DPP-read wait states without a single s_nop.
__device__ __forceinline__ void wave_allsum4(float &a, float &b, float &c, float &d) {
    float ra, rb, rc, rd;
#define ASG_DPP_STEP4(op, ctl) \
    op " %0, %0, %0 " ctl "\n" op " %1, %1, %1 " ctl "\n" op " %2, %2, %2 " ctl "\n" op " %3, %3, %3 " ctl "\n"
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        ASG_DPP_STEP4("v_add_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP4("v_add_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP4("v_add_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf")
        ASG_DPP_STEP4("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
        ASG_DPP_STEP4("v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1\n"
        : "=&v"(ra), "=&v"(rb), "=&v"(rc), "=&v"(rd) : "v"(a), "v"(b), "v"(c), "v"(d));
#undef ASG_DPP_STEP4
    a = readlane(ra, 63);
    b = readlane(rb, 63);
    c = readlane(rc, 63);
    d = readlane(rd, 63);
}
__device__ __forceinline__ void wave_allmax2(double &a, double &b);
__device__ __forceinline__ void wave_allsum2(double &a, double &b);

// fp64: plain DPP moves (correctness path, not tuned)
__device__ __forceinline__ double wave_allmax(double v) {
    v = fmax(v, dpp_mov<kDppXor1>(v, v));
    v = fmax(v, dpp_mov<kDppXor2>(v, v));
    v = fmax(v, dpp_mov<kDppHalfMirror>(v, v));
    v = fmax(v, dpp_mov<kDppMirror>(v, v));
    return fmax(fmax(readlane(v, 0), readlane(v, 16)), fmax(readlane(v, 32), readlane(v, 48)));
}
__device__ __forceinline__ double wave_allsum(double v) {
    v += dpp_mov<kDppXor1>(v, v);
    v += dpp_mov<kDppXor2>(v, v);
    v += dpp_mov<kDppHalfMirror>(v, v);
    v += dpp_mov<kDppMirror>(v, v);
    return (readlane(v, 0) + readlane(v, 16)) + (readlane(v, 32) + readlane(v, 48));
}

__device__ __forceinline__ void wave_allmax2(double &a, double &b) { a = wave_allmax(a); b = wave_allmax(b); }
__device__ __forceinline__ void wave_allsum2(double &a, double &b) { a = wave_allsum(a); b = wave_allsum(b); }

// log2(2^a + 2^b) for a, b finite or <= logzero (never NaN): max + log2(1 + 2^-(|a-b|))
template <typename R>
__device__ __forceinline__ R lse2(R a, R b) {
    R m = fmax(a, b);
    R d = fmin(a, b) - m;
    return m + Num<R>::log2(R(1) + Num<R>::exp2(d));
}

// ---- raw buffer stores: hardware bounds check instead of EXEC masking ------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
}
constexpr unsigned kOobOffset = 0x80000000u;   // voffset of lanes that must not store
#ifndef ASG_X_STORE_AUX
#define ASG_X_STORE_AUX 0      // developer A/B: cache policy bits of the plain fp32 state / row stores (17 = written through the L2)
#endif
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, ASG_X_STORE_AUX);
}
__device__ __forceinline__ void buf_store(double v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 w = {(unsigned) __double2loint(v), (unsigned) __double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, voff, soff, 0);
}

// two consecutive elements in one store (a scale-log entry)
__device__ __forceinline__ void buf_store2(V2<float> v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(u2{__float_as_uint(v.x), __float_as_uint(v.y)}, rs, voff, soff, 0);
}
__device__ __forceinline__ void buf_store2(V2<double> v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u4{(unsigned) __double2loint(v.x), (unsigned) __double2hiint(v.x),
                                              (unsigned) __double2loint(v.y), (unsigned) __double2hiint(v.y)}, rs, voff, soff, 0);
}

// raw buffer loads: per-lane byte offset in a VGPR, per-frame byte offset in an SGPR (one s_mul / s_add per load
// instead of a 64-bit address computation per lane)
template <typename R> __device__ __forceinline__ R buf_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float buf_load<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
template <> __device__ __forceinline__ double buf_load<double>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return __hiloint2double((int) w.y, (int) w.x);
}

// Per-frame scatter of the aligned posteriors (values in [0,1], at most 64 of them, sum <= 1 per label): the integer
// type of the deterministic LDS adds.  fp32 uses 32-bit words with a 2^-30 quantum (v_cvt_u32_f32 / ds_add_u32; the
// 64-bit route costs ~15 double-precision instructions per frame), fp64 keeps 64-bit words with a 2^-44 quantum.
template <typename R> struct FrameFix;
#ifdef ASG_OLD_FIX
template <> struct FrameFix<float> {
    typedef unsigned long long T;
    static __device__ __forceinline__ T to(float x) { return (T) __double2ll_rn((double) x * 1099511627776.0); }
    static __device__ __forceinline__ float from(T v) { return (float) ((double) (long long) v * (1.0 / 1099511627776.0)); }
};
#else
template <> struct FrameFix<float> {
    typedef unsigned T;
    static __device__ __forceinline__ T to(float x) { return (T) __builtin_rintf(x * 1073741824.0f); }
    static __device__ __forceinline__ float from(T v) { return (float) v * (1.0f / 1073741824.0f); }
};
#endif
template <> struct FrameFix<double> {
    typedef unsigned long long T;
    static __device__ __forceinline__ T to(double x) { return (T) __double2ll_rn(x * Num<double>::kFix); }
    static __device__ __forceinline__ double from(T v) { return (double) (long long) v * (1.0 / Num<double>::kFix); }
};

// fixed-point helpers for deterministic LDS scatter-adds (integer adds commute)
template <typename R> __device__ __forceinline__ unsigned long long to_fix(R x) {
    return (unsigned long long) __double2ll_rn((double) x * Num<R>::kFix);
}
template <typename R> __device__ __forceinline__ R from_fix(unsigned long long v) {
    return (R) ((double) (long long) v * (1.0 / Num<R>::kFix));
}

}  // namespace asg
