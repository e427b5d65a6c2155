// torch_asg_amd/csrc/asg_viterbi.hip -- best-path (Viterbi) force alignment on gfx950 (S <= 64: one wavefront per
// utterance; S <= 8192: one workgroup per utterance, up to eight positions per thread).
//
// The force-aligned lattice of /root/reference/torch_asg/native/force_aligned_lattice.cpp:84-111 in the tropical
// semiring (max instead of log-sum-exp: doc/tech_report.tex:84-88; a TODO in the reference's README.md:33 -- the
// reference has no implementation; the test suite pins the semantics by exhaustive path enumeration):
//   v[0][s] = I[0][O_0] for s == 0 else -inf;  v[t][s] = I[t][O_s] + max(v[t-1][s] + Tr[O_s][O_s], v[t-1][s-1] + Tr[O_s][O_{s-1}])
// One wavefront per utterance, lane s = target position.  Arithmetic is in the caller's natural-log units and in the
// same order as the plain-C restatement used by the tests (adds and compares only), so scores and paths are
// bit-identical to it.
// Forward: the neighbour value arrives through a wave_shr:1 DPP move; the 64 back-pointer bits of a frame are one
// v_cmp (ballot) and go to the workspace as one 8-byte store.  Backtrace: 64 frames at a time -- every lane loads one
// frame's mask, then the wave walks the 64 frames with v_readlane + scalar bit tests and writes the positions back
// with one coalesced store.
#include "asg_common.h"
#include "asg_kernels.h"

namespace asg {

namespace {

constexpr int kVPF = 16;      // emission prefetch depth (frames)

template <typename R>
__global__ void __launch_bounds__(64) viterbi_small_kernel(Problem P, unsigned long long *masks, R *scores, long long *path) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int T = P.T, S = P.S;
    const R NINF = Num<R>::ninf();
    int len = P.in_len ? (int) (P.in_len[b] < 0 ? 0 : (P.in_len[b] > T ? T : P.in_len[b])) : T;
    int ol = P.tg_len ? (int) (P.tg_len[b] < 0 ? 0 : (P.tg_len[b] > S ? S : P.tg_len[b])) : S;
    len = __builtin_amdgcn_readfirstlane(len);
    ol = __builtin_amdgcn_readfirstlane(ol);
    long long *pb = path + (long long) b * T;
    unsigned long long *mb = masks + (long long) b * T;
    // frames outside the utterance (and everything, if there is no alignment) read -1
    for (int t = lane; t < T; t += 64) pb[t] = -1;
    const bool feasible = len >= 1 && ol >= 1 && ol <= len;
    if (!feasible) {
        if (lane == 0) scores[b] = NINF;
        return;
    }
    const bool act = lane < ol;
    const int sc = act ? lane : 0, sp = (act && lane >= 1) ? lane - 1 : 0;
    const int64_t *tg = P.targets + (int64_t) b * P.gs0;
    const int64_t cur64 = tg[(int64_t) sc * P.gs1], prv64 = tg[(int64_t) sp * P.gs1];
    const int cur = (int) (cur64 < 0 ? 0 : (cur64 > P.N - 1 ? P.N - 1 : cur64));
    const int prv = (int) (prv64 < 0 ? 0 : (prv64 > P.N - 1 ? P.N - 1 : prv64));
    const R *tr = (const R *) P.transition;
    const R H = tr[(long long) cur * P.ts0 + (long long) cur * P.ts1];
    const R Dp = (act && lane >= 1) ? tr[(long long) cur * P.ts0 + (long long) prv * P.ts1] : NINF;
    const R *in = (const R *) P.inputs + (long long) b * P.is1 + (long long) cur * P.is2;

    R v = (lane == 0) ? in[0] : NINF;
    R ring[kVPF];
#pragma unroll
    for (int k = 0; k < kVPF; ++k) ring[k] = in[(long long) min(1 + k, len - 1) * P.is0];
    for (int t0 = 1; t0 < len; t0 += kVPF) {
        R nxt[kVPF];
#pragma unroll
        for (int k = 0; k < kVPF; ++k) nxt[k] = in[(long long) min(t0 + kVPF + k, len - 1) * P.is0];
#pragma unroll
        for (int k = 0; k < kVPF; ++k) {
            const int t = t0 + k;
            if (t < len) {
                const R stay = v + H;
                const R left = dpp_mov<kDppWaveShr1, true>(R(0), v);        // lane 0 reads 0; its Dp is -inf
                const R come = left + Dp;
                const bool take = come > stay;                               // ties keep "stay"
                const unsigned long long m = __ballot(take);
                const R em = act ? ring[k] : NINF;
                v = em + (take ? come : stay);
                if (lane == 0) mb[t] = m;
            }
        }
#pragma unroll
        for (int k = 0; k < kVPF; ++k) ring[k] = nxt[k];
    }
    const R best = readlane(v, ol - 1);
    if (!(best > NINF) || best != best) {          // no finite path (e.g. -inf emissions on every alignment)
        if (lane == 0) scores[b] = NINF;
        return;
    }
    if (lane == 0) scores[b] = best;
    // ---- backtrace: the masks were written by lane 0 of this wavefront; make them visible to all lanes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int s = ol - 1;
    for (int c0 = ((len - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
        const int t_mine = c0 + lane;
        unsigned long long mv = (t_mine >= 1 && t_mine < len) ? __builtin_nontemporal_load(mb + t_mine) : 0ull;
        const unsigned mlo = (unsigned) mv, mhi = (unsigned) (mv >> 32);
        int mine = -1;
        const int top = min(63, len - 1 - c0);
        for (int k = top; k >= 0; --k) {
            mine = (lane == k) ? s : mine;
            const unsigned lo = __builtin_amdgcn_readlane(mlo, k), hi = __builtin_amdgcn_readlane(mhi, k);
            const unsigned long long m = ((unsigned long long) hi << 32) | lo;
            s -= (int) ((m >> s) & 1ull);           // frame 0 has mask 0: s stays (it is 0 for a valid path)
        }
        if (t_mine < len) pb[t_mine] = mine;
    }
}

// S up to 1024 (KP = 1) / 4096 (KP = 4) / 8192 (KP = 8): one workgroup per utterance, thread tid owns target positions tid + 1024 k, 64 positions
// per mask word.  The left neighbour crosses thread boundaries through a double-buffered LDS line (one __syncthreads per
// frame); every 64-position strip stores its own 64 back-pointer bits per frame (masks[b][t][strip]).  The backtrace runs
// in wavefront 0, 64 frames at a time: within 64 frames the position moves by at most 64, so the two mask words a frame can
// need are loaded up front by its lane and the walk itself touches no memory.
template <typename R, int KP>
__global__ void __launch_bounds__(1024) viterbi_wide_kernel(Problem P, unsigned long long *masks, R *scores, long long *path) {
    __shared__ R line[2][1024 * KP + 1];
    __shared__ R best_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int T = P.T, S = P.S;
    const int nw = (S + 63) / 64;
    const R NINF = Num<R>::ninf();
    int len = P.in_len ? (int) (P.in_len[b] < 0 ? 0 : (P.in_len[b] > T ? T : P.in_len[b])) : T;
    int ol = P.tg_len ? (int) (P.tg_len[b] < 0 ? 0 : (P.tg_len[b] > S ? S : P.tg_len[b])) : S;
    long long *pb = path + (long long) b * T;
    unsigned long long *mb = masks + (long long) b * T * nw;
    for (int t = tid; t < T; t += blockDim.x) pb[t] = -1;
    const bool feasible = len >= 1 && ol >= 1 && ol <= len;
    if (!feasible) {
        if (tid == 0) scores[b] = NINF;
        return;
    }
    const int64_t *tg = P.targets + (int64_t) b * P.gs0;
    const R *tr = (const R *) P.transition;
    bool act[KP], in_s[KP];
    R H[KP], Dp[KP], v[KP];
    const R *in[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int p = tid + 1024 * k;
        act[k] = p < ol;
        in_s[k] = p < nw * 64;                           // (its strip has a mask word)
        const int sc = act[k] ? p : 0, sp = (act[k] && p >= 1) ? p - 1 : 0;
        const int64_t cur64 = tg[(int64_t) sc * P.gs1], prv64 = tg[(int64_t) sp * P.gs1];
        const int cur = (int) (cur64 < 0 ? 0 : (cur64 > P.N - 1 ? P.N - 1 : cur64));
        const int prv = (int) (prv64 < 0 ? 0 : (prv64 > P.N - 1 ? P.N - 1 : prv64));
        H[k] = tr[(long long) cur * P.ts0 + (long long) cur * P.ts1];
        Dp[k] = (act[k] && p >= 1) ? tr[(long long) cur * P.ts0 + (long long) prv * P.ts1] : NINF;
        in[k] = (const R *) P.inputs + (long long) b * P.is1 + (long long) cur * P.is2;
        v[k] = (p == 0) ? in[k][0] : NINF;
    }
    if (tid == 0) { line[0][0] = NINF; line[1][0] = NINF; }     // the neighbour of position 0
    constexpr int PF = KP == 1 ? 8 : 2;
    R ring[PF][KP];
#pragma unroll
    for (int q = 0; q < PF; ++q)
#pragma unroll
        for (int k = 0; k < KP; ++k) ring[q][k] = in[k][(long long) min(1 + q, len - 1) * P.is0];
    for (int t0 = 1; t0 < len; t0 += PF) {
        R nxt[PF][KP];
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int k = 0; k < KP; ++k) nxt[q][k] = in[k][(long long) min(t0 + PF + q, len - 1) * P.is0];
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int t = t0 + q;
            if (t < len) {                                       // uniform
                R *ln = line[t & 1];
#pragma unroll
                for (int k = 0; k < KP; ++k) ln[tid + 1024 * k + 1] = v[k];
                __syncthreads();
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const R stay = v[k] + H[k];
                    const R come = ln[tid + 1024 * k] + Dp[k];
                    const bool take = come > stay;
                    const unsigned long long m = __ballot(take);
                    const R em = act[k] ? ring[q][k] : NINF;
                    v[k] = em + (take ? come : stay);
                    if (lane == 0 && in_s[k]) mb[(long long) t * nw + wave + 16 * k] = m;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int k = 0; k < KP; ++k) ring[q][k] = nxt[q][k];
    }
#pragma unroll
    for (int k = 0; k < KP; ++k)
        if (tid + 1024 * k == ol - 1) best_sh = v[k];
    __threadfence_block();
    __syncthreads();                      // also orders the mask stores of all wavefronts before the loads below
    const R best = best_sh;
    if (!(best > NINF) || best != best) {
        if (tid == 0) scores[b] = NINF;
        return;
    }
    if (tid == 0) scores[b] = best;
    if (wave != 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int s = ol - 1;
    for (int c0 = ((len - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
        const int t_mine = c0 + lane;
        const int w_hi = s >> 6, w_lo = max(w_hi - 1, 0);
        unsigned long long mh = 0, ml = 0;
        if (t_mine >= 1 && t_mine < len) {
            mh = __builtin_nontemporal_load(mb + (long long) t_mine * nw + w_hi);
            ml = __builtin_nontemporal_load(mb + (long long) t_mine * nw + w_lo);
        }
        int mine = -1;
        const int top = min(63, len - 1 - c0);
        for (int k = top; k >= 0; --k) {
            mine = (lane == k) ? s : mine;
            const bool hi = (s >> 6) == w_hi;
            const unsigned lo32 = __builtin_amdgcn_readlane(hi ? (unsigned) mh : (unsigned) ml, k);
            const unsigned hi32 = __builtin_amdgcn_readlane(hi ? (unsigned) (mh >> 32) : (unsigned) (ml >> 32), k);
            const unsigned long long m = ((unsigned long long) hi32 << 32) | lo32;
            s -= (int) ((m >> (s & 63)) & 1ull);
        }
        if (t_mine < len) pb[t_mine] = mine;
    }
}

}  // namespace

template <typename R>
hipError_t launch_viterbi_small(const Problem &P, void *work, void *scores, void *path, hipStream_t stream) {
    if (P.S <= 64)
        hipLaunchKernelGGL((viterbi_small_kernel<R>), dim3(P.B), dim3(64), 0, stream, P, (unsigned long long *) work,
                           (R *) scores, (long long *) path);
    else if (P.S <= 1024)
        hipLaunchKernelGGL((viterbi_wide_kernel<R, 1>), dim3(P.B), dim3((P.S + 63) / 64 * 64), 0, stream, P,
                           (unsigned long long *) work, (R *) scores, (long long *) path);
    else if (P.S <= 4096)
        hipLaunchKernelGGL((viterbi_wide_kernel<R, 4>), dim3(P.B), dim3(1024), 0, stream, P,
                           (unsigned long long *) work, (R *) scores, (long long *) path);
    else if (P.S <= 8192)
        hipLaunchKernelGGL((viterbi_wide_kernel<R, 8>), dim3(P.B), dim3(1024), 0, stream, P,
                           (unsigned long long *) work, (R *) scores, (long long *) path);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
template hipError_t launch_viterbi_small<float>(const Problem &, void *, void *, void *, hipStream_t);
template hipError_t launch_viterbi_small<double>(const Problem &, void *, void *, void *, hipStream_t);

}  // namespace asg
