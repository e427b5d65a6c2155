// torch_asg_amd/csrc/asg_generic_aligned.hip -- generic path, FORCE-ALIGNED LATTICE with long targets (64 < S <= 8192) over any
// alphabet: the forward kernels (one wavefront with K positions per lane / pipelined / a barrier per frame / strips through LDS) and the
// gradient kernels with their label and edge scatters.  Restates force_aligned_lattice.cpp:84-264.  The file-level story is in asg_generic.hip.
#include "asg_generic_common.h"

namespace asg {

namespace {

// Stored states of the aligned lattice on the long-target routes (S > 64) are DOUBLES, whatever the problem's precision.
// A state is stored relative to ONE reference per frame and direction (the direction's largest state, or an extrapolation
// of it); with transition scores of tens of nats and slack between target and input length the state ON the dominant path
// can sit ~1000 log2 units below that reference while the other direction's state compensates, and 1000 costs a float
// 6e-5: the posterior exp2(ab + bb - max) was off by up to 1.2e-4 (round 3: tools/fuzz_routes.py).  The recursions carry
// their states in double anyway; storing them unrounded costs S * T * B * 8 more bytes of traffic and holds 1e-4.
// (S <= 64: the one-wavefront chains of asg_chains.h store the problem's type; tools/stress_duo.py bounds them at 7e-5.)
typedef double AlignedState;
template <typename SR> __device__ __forceinline__ double load_state(const SR *p, int64_t i) { return (double) p[i]; }

// ------------------------------------------------------------------ aligned lattice, long targets (64 < S <= 512)
// grid = (B, 2), block = 64: ONE wavefront per chain, lane l owns the K CONSECUTIVE target positions K l .. K l + K - 1
// (K = 2, 4, 8), so all but one neighbour of a frame's update sit in the lane's own registers and the last one comes
// from the lane next door by DPP -- no LDS, no barrier (aligned_wide_kernel below pays one workgroup barrier per frame:
// 428 ns per frame at S = 200 against ~120 here).  Same stored states, scores and side tables as aligned_wide_kernel;
// force_aligned_lattice.cpp:84-154 is what both restate.
template <typename R, int K, bool STORE>
__global__ void __launch_bounds__(64) aligned_long_kernel(Problem P, State W, FwdOut O, int mask) {
    constexpr int PF = K <= 4 ? 4 : 2;          // frames of emissions in flight ahead of the recursion
    const int b = blockIdx.x;
    const bool beta = (mask == kAlignedBeta) || (mask == (kAlignedAlpha | kAlignedBeta) && blockIdx.y == 1);
    const int lane = threadIdx.x, S = P.S, T = P.T, N = P.N;
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const int64_t *tg = P.targets + (int64_t) b * P.gs0;
    const R *tr = (const R *) P.transition;
    const R *inb = (const R *) P.inputs + (int64_t) b * P.is1;
    bool act[K];
    double ebias[K];                            // 0 on positions inside the target, log-zero otherwise: em = raw * log2 e + ebias
    unsigned eoff[K], soffv[K];                 // byte offsets: the label's emission inside a frame row; the position inside a state row
    double H2[K], Dx[K];                        // stay edge; alpha: edge from the previous position, beta: edge to the next
    // frames through buffer accesses: lane offset in a VGPR, frame offset in an SGPR (launch_fwd_generic checks that both
    // fit 32 bits); stores of positions >= S go out of bounds = nowhere (no EXEC juggling per position)
    __amdgpu_buffer_rsrc_t rin = make_rsrc((R *) inb, 0xffffffffu);
    __amdgpu_buffer_rsrc_t rout = make_rsrc((AlignedState *) (beta ? W.bb : W.ab) + (int64_t) b * T * S, STORE ? (unsigned) ((int64_t) T * S * sizeof(AlignedState)) : 0u);
    const unsigned frame_bytes = (unsigned) P.is0 * (unsigned) sizeof(R), row_bytes = (unsigned) S * (unsigned) sizeof(AlignedState);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = lane * K + k;
        act[k] = p < ol;
        const int cur = act[k] ? gclampi(tg[(int64_t) p * P.gs1], 0, N - 1) : 0;
        const int prv = (act[k] && p >= 1) ? gclampi(tg[(int64_t) (p - 1) * P.gs1], 0, N - 1) : 0;
        const int nxt = (p + 1 < ol) ? gclampi(tg[(int64_t) (p + 1) * P.gs1], 0, N - 1) : 0;
        const R h2 = act[k] ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : R(0);
        const R dp = (act[k] && p >= 1) ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) prv * P.ts1] * L2E, LZ) : LZ;
        const R dn = (p + 1 < ol) ? fmax(tr[(int64_t) nxt * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : LZ;
        H2[k] = (double) h2;
        Dx[k] = (double) (beta ? dn : dp);
        ebias[k] = act[k] ? 0.0 : -1e30;
        eoff[k] = (unsigned) (cur * (int) P.is2) * (unsigned) sizeof(R);
        soffv[k] = p < S ? (unsigned) p * (unsigned) sizeof(AlignedState) : kOobOffset;
        if (STORE && !beta && p < S) {
            V2<R> u = {h2, dp};
            reinterpret_cast<V2<R> *>(W.asu)[(int64_t) b * S + p] = u;
            int2 ii = {cur, prv};
            reinterpret_cast<int2 *>(W.asi)[(int64_t) b * S + p] = ii;
        }
    }
    R *score_out = (R *) (beta ? O.aligned_scores : O.aligned_scores_alpha);
    if (len < 1 || ol < 1) {
        if (lane == 0 && score_out) score_out[b] = Num<R>::ninf();
        return;
    }
    const double kZ = -1e30, L2Ed = 1.4426950408889634;
    // log2(2^x + 2^y) = max + log2(1 + 2^-|x - y|): the difference in double, the correction term in the problem's precision (the
    // modulus and the sign are source modifiers of the conversion and of v_exp).  Inside a step only the emission term is clamped at
    // log zero (-1e30; a -inf emission must not put -inf into a state: two of them side by side are inf - inf); a state can fall
    // below it by one -1e30 per frame until the next renormalisation (every 4 frames) clamps it -- nowhere near the range of a
    // double -- and the stores clamp what they write.
    auto lse2d = [&](double x, double y) {
        const R d = (R) fabs(x - y);
        return fmax(x, y) + (double) Num<R>::log2(R(1) + Num<R>::exp2(-d));
    };
    auto st = [&](double x) { return (AlignedState) fmax(x, kZ); };
    auto store_row = [&](int t, const double (&v)[K]) {
        if (!STORE) return;
        const unsigned so = (unsigned) __builtin_amdgcn_readfirstlane(t) * row_bytes;
#pragma unroll
        for (int k = 0; k < K; ++k) buf_store(st(v[k]), rout, soffv[k], so);
    };
    // frame f's emissions of this lane's labels (clamped frame index: the surplus loads of the last block are never used)
    auto fetch = [&](int f, R (&e)[K]) {
        const unsigned so = (unsigned) __builtin_amdgcn_readfirstlane(gclampi(f, 0, len - 1)) * frame_bytes;
#pragma unroll
        for (int k = 0; k < K; ++k) e[k] = buf_load<R>(rin, eoff[k], so);
    };
    double C = 0.0, v[K];
    auto renorm = [&]() {
        double mx = v[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmax(mx, v[k]);
        const R m = wave_allmax((R) mx);
        if (m > R(-1e29)) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = fmax(v[k] - (double) m, kZ);
            C += (double) m;
        }
    };
    R cur[PF][K], nxt[PF][K];
    // the PF K emission loads of the NEXT block are issued before this block's PF K state stores: "at most PF K memory
    // operations outstanding" = they have landed.  Said explicitly (left alone hipcc drains the store queue, vmcnt(0), at
    // every use of a loaded value: the previous frame's stores, every frame)
    constexpr int kOut = PF * K;
    constexpr int kWaitLoads = STORE ? (((kOut >> 4) << 14) | 0x0F70 | (kOut & 15)) : 0x0F70;
    if (!beta) {
        {
            R e0[K];
            fetch(0, e0);
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = (lane == 0 && k == 0 && act[0]) ? fmax((double) e0[0] * L2Ed, kZ) : kZ;
        }
        store_row(0, v);
#pragma unroll
        for (int u = 0; u < PF; ++u) fetch(1 + u, cur[u]);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int t0 = 1; t0 < len; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) fetch(t0 + PF + u, nxt[u]);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                if (t < len) {
                    // (everything below is branch-free: an `act ? .. : ..` around the transcendentals becomes an EXEC-masked
                    // branch per position, which also keeps the K independent updates from overlapping.  Lane 0's left
                    // neighbour reads 0, and position 0 has a log-zero arrive edge: no select needed)
                    const double left = prev_lane_or_zero<double>(v[K - 1]);
#pragma unroll
                    for (int k = K - 1; k >= 0; --k) {
                        const double em = fmax(fma((double) cur[u][k], L2Ed, ebias[k]), kZ);      // (a -inf emission stays finite)
                        const double from = k == 0 ? left : v[k - 1];
                        v[k] = em + lse2d(v[k] + H2[k], from + Dx[k]);
                    }
                    if ((t & 3) == 0) renorm();       // (every 4 frames: the stored floats stay within a few frames' growth of the offset)
                    store_row(t, v);
                }
            }
            __builtin_amdgcn_s_waitcnt(kWaitLoads);
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int k = 0; k < K; ++k) cur[u][k] = nxt[u][k];
        }
        if (score_out) {
            const int pl = ol - 1;
            double mine = v[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mine = (pl % K == k) ? v[k] : mine;
            const double last = __shfl(mine, pl / K);
            if (lane == 0) {
                const double sc = C + last;
                score_out[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = (lane * K + k == ol - 1) ? 0.0 : kZ;
        store_row(len - 1, v);
        // step u of a block that starts at frame t0 consumes the emissions of frame t0 - u and writes frame t0 - u - 1
#pragma unroll
        for (int u = 0; u < PF; ++u) fetch(len - 1 - u, cur[u]);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int t0 = len - 1; t0 >= 1; t0 -= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) fetch(t0 - PF - u, nxt[u]);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 - u;
                if (t >= 1) {
                    double y[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) y[k] = fmax(fma((double) cur[u][k], L2Ed, ebias[k]), kZ) + v[k];
                    const double right = next_lane_or_zero<double>(y[0]);      // (lane 63 reads 0; its last position has a log-zero leave edge)
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const double to = k == K - 1 ? right : y[k + 1];
                        v[k] = lse2d(y[k] + H2[k], to + Dx[k]);
                    }
                    if ((t & 3) == 0) renorm();       // (every 4 frames: the stored floats stay within a few frames' growth of the offset)
                    store_row(t - 1, v);
                }
            }
            __builtin_amdgcn_s_waitcnt(kWaitLoads);
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
                for (int k = 0; k < K; ++k) cur[u][k] = nxt[u][k];
        }
        if (score_out) {
            R e0[K];
            fetch(0, e0);
            if (lane == 0) {
                const double em = act[0] ? (double) e0[0] * L2Ed : kZ;
                const double sc = C + (em + v[0]);
                score_out[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
            }
        }
    }
}


// ------------------------------------------------------------------ aligned lattice, long targets, pipelined wavefronts
// grid = (B, 2), block = 64 * ceil(S / 64) (<= 1024): thread p owns target position p (one position per lane: the
// per-position work of a frame is a dozen double-precision / transcendental instructions, and one SIMD retires them at
// ~8 cycles apiece -- K positions per lane cost K times that, aligned_long_kernel: 276 us at S = 200, T = 1000).  The
// wavefronts of a chain form a PIPELINE instead of meeting at a workgroup barrier every frame (aligned_wide_kernel):
// the only value that crosses a wavefront boundary per frame (alpha: the state of position 64 w - 1, beta: y of position
// 64 (w + 1)) travels through a 64-slot LDS ring as a (value, frame) pair, the consumer polls the frame tag.  All
// wavefronts of a workgroup are co-resident, the dependency runs one way, so nothing can dead-lock.
// States are kept ABSOLUTE in double (no renormalisation inside the recursion); what is stored for the gradient pass is
// float(v - Cref), Cref = the largest state two 16-frame blocks ago, gathered once per block through LDS -- the same
// per-frame offset for every position of a frame, which is all the gradient pass needs.  Waiting for that gather also
// bounds the skew between the fastest and the slowest wavefront to two blocks, which is what makes 64 ring slots enough.
template <typename R, bool STORE>
__global__ void __launch_bounds__(1024) aligned_pipe_kernel(Problem P, State W, FwdOut O, int mask) {
    constexpr int D = 64;
    typedef int I4 __attribute__((ext_vector_type(4)));
    // one 16-byte slot per (wavefront, frame mod D): {value lo, frame, value hi, frame} -- written with ONE ds_write_b128 and
    // read with one ds_read_b128; the frame tag sits in both 8-byte halves, so a reader that finds it in both has the value
    __shared__ __attribute__((aligned(16))) I4 ring[16][D];
    __shared__ float blk_m[4][16];
    __shared__ int blk_t[4][16];
    const int b = blockIdx.x;
    const bool beta = (mask == kAlignedBeta) || (mask == (kAlignedAlpha | kAlignedBeta) && blockIdx.y == 1);
    const int s = threadIdx.x, lane = s & 63, wave = __builtin_amdgcn_readfirstlane(s >> 6), NW = (int) (blockDim.x >> 6);
    const int S = P.S, T = P.T, N = P.N;
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const bool act = s < ol;
    const int64_t *tg = P.targets + (int64_t) b * P.gs0;
    const int cur = act ? gclampi(tg[(int64_t) s * P.gs1], 0, N - 1) : 0;
    const int prv = (act && s >= 1) ? gclampi(tg[(int64_t) (s - 1) * P.gs1], 0, N - 1) : 0;
    const int nxt = (s + 1 < ol) ? gclampi(tg[(int64_t) (s + 1) * P.gs1], 0, N - 1) : 0;
    const R *tr = (const R *) P.transition;
    const R H2f = act ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : R(0);
    const R Dprev = (act && s >= 1) ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) prv * P.ts1] * L2E, LZ) : LZ;
    const R Dnext = (s + 1 < ol) ? fmax(tr[(int64_t) nxt * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : LZ;
    if (STORE && !beta && s < S) {
        V2<R> u = {H2f, Dprev};
        reinterpret_cast<V2<R> *>(W.asu)[(int64_t) b * S + s] = u;
        int2 ii = {cur, prv};
        reinterpret_cast<int2 *>(W.asi)[(int64_t) b * S + s] = ii;
    }
    R *score_out = (R *) (beta ? O.aligned_scores : O.aligned_scores_alpha);
    if (len < 1 || ol < 1) {
        if (s == 0 && score_out) score_out[b] = Num<R>::ninf();
        return;
    }
    for (int q = s; q < 16 * D; q += (int) blockDim.x) (&ring[0][0])[q] = I4{0, -1, 0, -1};
    if (s < 64) (&blk_t[0][0])[s] = -1;
    __syncthreads();
    const double kZ = -1e30, L2Ed = 1.4426950408889634;
    const double H2 = (double) H2f, Dx = (double) (beta ? Dnext : Dprev), ebias = act ? 0.0 : -1e30;
    auto lse2d = [&](double x, double y) {
        const double m = fmax(x, y);
        const R d = (R) (fmin(x, y) - m);
        return m + (double) Num<R>::log2(R(1) + Num<R>::exp2(d));
    };
    // emissions of this position's label: frame offset in an SGPR (32-bit offsets checked by the launcher)
    __amdgpu_buffer_rsrc_t rin = make_rsrc((R *) P.inputs + (int64_t) b * P.is1, 0xffffffffu);
    const unsigned eoff = (unsigned) (cur * (int) P.is2) * (unsigned) sizeof(R), frame_bytes = (unsigned) P.is0 * (unsigned) sizeof(R);
    auto emis = [&](int f) -> R {
        return buf_load<R>(rin, eoff, (unsigned) __builtin_amdgcn_readfirstlane(gclampi(f, 0, len - 1)) * frame_bytes);
    };
    __amdgpu_buffer_rsrc_t rout = make_rsrc((AlignedState *) (beta ? W.bb : W.ab) + (int64_t) b * T * S, STORE ? (unsigned) ((int64_t) T * S * sizeof(AlignedState)) : 0u);
    const unsigned soff = s < S ? (unsigned) s * (unsigned) sizeof(AlignedState) : kOobOffset, row_bytes = (unsigned) S * (unsigned) sizeof(AlignedState);
    // reference offset of the stored states: frame-0 emission of the first target label to start with (every thread can
    // compute it), then the block maxima
    double Cref;
    {
        const int c0 = gclampi(tg[0], 0, N - 1);
        const R e0 = ((const R *) P.inputs)[(int64_t) b * P.is1 + (int64_t) (beta ? len - 1 : 0) * P.is0 + (int64_t) c0 * P.is2];
        Cref = beta ? 0.0 : (double) e0 * L2Ed;
    }
    // ... extrapolated linearly: the largest state two blocks ago plus the growth per step between the last two gathered
    // maxima (transition scores of tens of nats move the scores by ~50 log2 units per frame: 32 frames of that above a
    // constant reference would cost the stored floats 1e-4 of precision).  Every wavefront derives the same numbers.
    double Cslope = 0.0;
    int Cstep = 0;                                   // the step Cref belongs to
    bool Chave = false;
    auto store = [&](int t, int n, double v) {       // frame t, step n
        if (STORE) buf_store((AlignedState) fmax(v - fma(Cslope, (double) (n - Cstep), Cref), kZ), rout, soff,
                             (unsigned) __builtin_amdgcn_readfirstlane(t) * row_bytes);
    };
    // step n = 1, 2, ...: block boundary bookkeeping.  End of block k (n & 15 == 15): publish this wavefront's largest
    // state; start of block k >= 2: Cref = max over the wavefronts of their block k - 2 maxima.
    auto block_end = [&](int n, double v) {
        const float m = wave_allmax((float) fmax(v, kZ));
        if (lane == 0) {
            const int k = n >> 4;
            // (LDS executes one wavefront's accesses in order: value first, tag second is all the ordering a reader of the
            // tag needs.  A RELEASE store would also wait for this wavefront's pending global stores -- the frame's
            // state -- every frame: 700+ cycles per step instead of ~250)
            blk_m[k & 3][wave] = m;
            asm volatile("" ::: "memory");
            __hip_atomic_store(&blk_t[k & 3][wave], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto block_begin = [&](int n) {
        const int k = (n >> 4) - 2;
        if (k < 0) return;
        float m = -3e38f;
        for (int w = 0; w < NW; ++w) {
            while (__hip_atomic_load(&blk_t[k & 3][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != k) __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
            m = fmaxf(m, __hip_atomic_load(&blk_m[k & 3][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
        if (m > -1e29f) {
            const int at = 16 * k + 15;
            Cslope = Chave ? ((double) m - Cref) / (double) (at - Cstep) : 0.0;
            Cref = (double) m;
            Cstep = at;
            Chave = true;
        }
    };
    // the slot wavefront `w` uses for step `n` (16 bytes, one LDS read; valid when both tags say n)
    auto peek = [&](int w, int n) -> I4 {
        // (not `volatile`: address-space inference skips volatile accesses and this would become a flat load; the empty asm
        // keeps the compiler from caching or hoisting the read)
        asm volatile("" ::: "memory");
        const I4 r = ring[w][n & (D - 1)];
        asm volatile("" ::: "memory");
        return r;
    };
    // the value of step `n` from a slot read earlier (`raw`: normally the look-ahead read of the previous frame, long landed);
    // re-read until the producer has written it
    auto take = [&](int w, int n, I4 raw) -> double {
        while (raw.y != n || raw.w != n) {
            __builtin_amdgcn_s_sleep(1);
            raw = peek(w, n);
        }
        return __hiloint2double(raw.z, raw.x);
    };
    auto give = [&](int n, double v) {             // (by ONE lane of this wavefront)
        const I4 pk = {__double2loint(v), n, __double2hiint(v), n};
        asm volatile("" ::: "memory");
        ring[wave][n & (D - 1)] = pk;
        asm volatile("" ::: "memory");
    };
    // Emissions are fetched a 16-frame block ahead: the loads of the next block are issued before this block's 16 state
    // stores, so "at most 16 memory operations outstanding" means they have all landed -- said explicitly below; left to
    // itself hipcc waits for vmcnt(0) at every use of a loaded value, i.e. for the previous frame's store, every frame
    // (700 cycles per frame instead of ~250).
    constexpr int PF = 16;
    R ecur[PF], enxt[PF];
    double v;
    if (!beta) {
        v = (s == 0) ? fma((double) emis(0), L2Ed, ebias) : kZ;
        store(0, 0, v);
        if (wave < NW - 1 && lane == 63) give(0, v);
#pragma unroll
        for (int u = 0; u < PF; ++u) ecur[u] = emis(1 + u);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        I4 ahead = {0, -1, 0, -1};                   // the neighbour's slot of the NEXT frame, read one frame early
        if (wave > 0) ahead = peek(wave - 1, 0);
        for (int t0 = 1; t0 < len; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) enxt[u] = emis(t0 + PF + u);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;                       // step n = t
                if (t < len) {
                    if ((t & 15) == 0) block_begin(t);
                    double left = prev_lane_or_zero<double>(v);
                    if (wave > 0) {
                        const double nb = take(wave - 1, t - 1, ahead);
                        ahead = peek(wave - 1, t);
                        left = lane == 0 ? nb : left;
                    }
                    const double em = fma((double) ecur[u], L2Ed, ebias);
                    v = em + lse2d(v + H2, left + Dx);
                    if (wave < NW - 1 && lane == 63) give(t, v);
                    store(t, t, v);
                    if ((t & 15) == 15) block_end(t, v);
                }
            }
            __builtin_amdgcn_s_waitcnt(STORE ? 0x4F70 : 0x0F70);
#pragma unroll
            for (int u = 0; u < PF; ++u) ecur[u] = enxt[u];
        }
        if (score_out) {
            if (s == ol - 1) score_out[b] = (v < -1e29) ? Num<R>::ninf() : (R) (v * kLn2);
        }
    } else {
        v = (s == ol - 1) ? 0.0 : kZ;
        store(len - 1, 0, v);
#pragma unroll
        for (int u = 0; u < PF; ++u) ecur[u] = emis(len - 1 - u);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // step n = len - t (1, 2, ...) consumes the emissions of frame t and writes frame t - 1
        I4 ahead = {0, -1, 0, -1};
        if (wave < NW - 1) ahead = peek(wave + 1, 1);
        for (int t0 = len - 1; t0 >= 1; t0 -= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) enxt[u] = emis(t0 - PF - u);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 - u, n = len - t;
                if (t >= 1) {
                    if ((n & 15) == 0) block_begin(n);
                    const double y = fma((double) ecur[u], L2Ed, ebias) + v;
                    if (wave > 0 && lane == 0) give(n, y);
                    double right = next_lane_or_zero<double>(y);
                    if (wave < NW - 1) {
                        const double nb = take(wave + 1, n, ahead);
                        ahead = peek(wave + 1, n + 1);
                        right = lane == 63 ? nb : right;
                    }
                    v = lse2d(y + H2, right + Dx);
                    store(t - 1, n, v);
                    if ((n & 15) == 15) block_end(n, v);
                }
            }
            __builtin_amdgcn_s_waitcnt(STORE ? 0x4F70 : 0x0F70);
#pragma unroll
            for (int u = 0; u < PF; ++u) ecur[u] = enxt[u];
        }
        if (score_out) {
            if (s == 0) {
                const double sc = fma((double) emis(0), L2Ed, ebias) + v;
                score_out[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
            }
        }
    }
}

// ------------------------------------------------------------------ aligned lattice, wide targets
// grid = (B, 2), block = 64 * ceil(S/64) (<= 1024).  blockIdx.y: 0 = alpha, 1 = beta.  Thread s owns target
// position s; the neighbour's value travels through a double-buffered LDS row (one barrier per frame).
template <typename R, bool STORE>
__global__ void __launch_bounds__(1024) aligned_wide_kernel(Problem P, State W, FwdOut O, int mask) {
    __shared__ double row[2][1024 + 2];
    __shared__ R red[16];
    const int b = blockIdx.x;
    const bool beta = (mask == kAlignedBeta) || (mask == (kAlignedAlpha | kAlignedBeta) && blockIdx.y == 1);
    const int s = threadIdx.x, S = P.S, T = P.T, N = P.N;
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const bool act = s < ol;
    const int64_t *tg = P.targets + (int64_t) b * P.gs0;
    const int cur = act ? gclampi(tg[(int64_t) s * P.gs1], 0, N - 1) : 0;
    const int prv = (act && s >= 1) ? gclampi(tg[(int64_t) (s - 1) * P.gs1], 0, N - 1) : 0;
    const int nxt = (s + 1 < ol) ? gclampi(tg[(int64_t) (s + 1) * P.gs1], 0, N - 1) : 0;
    const R *tr = (const R *) P.transition;
    const R H2 = act ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : R(0);
    const R Dprev = (act && s >= 1) ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) prv * P.ts1] * L2E, LZ) : LZ;
    const R Dnext = (s + 1 < ol) ? fmax(tr[(int64_t) nxt * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : LZ;
    const R *in = (const R *) P.inputs + (int64_t) b * P.is1 + (int64_t) cur * P.is2;
    AlignedState *out = (AlignedState *) (beta ? W.bb : W.ab) + (int64_t) b * T * S;
    if (STORE && !beta && s < S) {
        V2<R> u = {H2, Dprev};
        reinterpret_cast<V2<R> *>(W.asu)[(int64_t) b * S + s] = u;
        int2 ii = {cur, prv};
        reinterpret_cast<int2 *>(W.asi)[(int64_t) b * S + s] = ii;
    }
    R *score_out = (R *) (beta ? O.aligned_scores : O.aligned_scores_alpha);
    if (len < 1 || ol < 1) {
        if (s == 0 && score_out) score_out[b] = Num<R>::ninf();
        return;
    }
    // running state in double (see asg_small.hip: an fp32 log-domain state loses ~2e-6 per frame at off-peak
    // positions); only the bounded correction log2(1 + 2^d) is evaluated in the problem's precision
    const double kZ = -1e30, L2Ed = 1.4426950408889634;
    auto lse2d = [&](double x, double y) {
        const double m = fmax(x, y);
        const R d = (R) (fmin(x, y) - m);
        return m + (double) Num<R>::log2(R(1) + Num<R>::exp2(d));
    };
    auto st = [&](double x) { return (AlignedState) fmax(x, kZ); };
    double C = 0.0;
    double v;
    if (!beta) {
        v = (s == 0) ? fmax((double) in[0] * L2Ed, kZ) : kZ;
        if (!act) v = kZ;
        if (STORE && s < S) out[s] = st(v);
        for (int t = 1; t < len; ++t) {
            double *rw = row[t & 1];
            rw[s + 1] = v;
            if (s == 0) rw[0] = kZ;
            __syncthreads();
            const double em = act ? (double) in[(int64_t) t * P.is0] * L2Ed : kZ;
            const double left = rw[s];
            v = fmax(em + lse2d(v + (double) H2, left + (double) Dprev), kZ);
            if ((t & 15) == 0) {            // renormalise now and then: log domain is offset free
                R m = wave_allmax((R) v);
                if ((s & 63) == 0) red[s >> 6] = m;
                __syncthreads();
                R mm = red[0];
                for (int w = 1; w < (int) (blockDim.x >> 6); ++w) mm = fmax(mm, red[w]);
                if (mm > R(-1e29)) { v = fmax(v - (double) mm, kZ); C += (double) mm; }
                __syncthreads();
            }
            if (STORE && s < S) out[(int64_t) t * S + s] = st(v);
        }
        if (score_out) {
            row[0][s] = v;
            __syncthreads();
            if (s == 0) {
                double sc = C + row[0][ol - 1];
                score_out[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
            }
        }
    } else {
        v = (s == ol - 1) ? 0.0 : kZ;
        if (STORE && s < S) out[(int64_t) (len - 1) * S + s] = st(v);
        for (int t = len - 1; t >= 1; --t) {
            const double em = act ? (double) in[(int64_t) t * P.is0] * L2Ed : kZ;
            const double y = fmax(em + v, kZ);
            double *rw = row[t & 1];
            rw[s] = y;
            if (s == (int) blockDim.x - 1) rw[blockDim.x] = kZ;
            __syncthreads();
            const double right = rw[s + 1];
            v = fmax(lse2d(y + (double) H2, right + (double) Dnext), kZ);
            if ((t & 15) == 0) {
                R m = wave_allmax((R) v);
                if ((s & 63) == 0) red[s >> 6] = m;
                __syncthreads();
                R mm = red[0];
                for (int w = 1; w < (int) (blockDim.x >> 6); ++w) mm = fmax(mm, red[w]);
                if (mm > R(-1e29)) { v = fmax(v - (double) mm, kZ); C += (double) mm; }
                __syncthreads();
            }
            if (STORE && s < S) out[(int64_t) (t - 1) * S + s] = st(v);
        }
        if (score_out) {
            const double em = act ? (double) in[0] * L2Ed : kZ;
            if (s == 0) {
                double sc = C + (em + v);
                score_out[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
            }
        }
    }
}

// ------------------------------------------------------------------ aligned lattice, very long targets (1024 < S <= 8192)
constexpr int kMaxTargets = 8192;
// The reference takes any target length (force_aligned_lattice.cpp:84-154 has no limit); the kernels above stop at one
// position per thread of a 1024-thread workgroup.  Beyond that the same recursion is strip-mined: thread s owns positions
// s, s + 1024, ... (KP of them), the whole frame's states travel through a double-buffered LDS row (one barrier per frame, as
// aligned_wide_kernel).  A correctness route, not a tuned one: targets of thousands of positions are hours of audio.
// grid = (B, 2), block = 1024, dynamic LDS = 2 (S + 2) doubles.
template <typename R, bool STORE, int KP>
__global__ void __launch_bounds__(1024) aligned_strip_kernel(Problem P, State W, FwdOut O, int mask) {
    extern __shared__ __attribute__((aligned(16))) double strip_row[];      // [2][S + 2]
    __shared__ R red[16];
    const int b = blockIdx.x;
    const bool beta = (mask == kAlignedBeta) || (mask == (kAlignedAlpha | kAlignedBeta) && blockIdx.y == 1);
    const int tid = threadIdx.x, S = P.S, T = P.T, N = P.N;
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const int64_t *tg = P.targets + (int64_t) b * P.gs0;
    const R *tr = (const R *) P.transition;
    double *row0 = strip_row, *row1 = strip_row + (S + 2);
    bool act[KP];
    double H2[KP], Dx[KP];
    const R *in[KP];
    AlignedState *out = (AlignedState *) (beta ? W.bb : W.ab) + (int64_t) b * T * S;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int p = tid + 1024 * k;
        act[k] = p < ol;
        const int cur = act[k] ? gclampi(tg[(int64_t) p * P.gs1], 0, N - 1) : 0;
        const int prv = (act[k] && p >= 1) ? gclampi(tg[(int64_t) (p - 1) * P.gs1], 0, N - 1) : 0;
        const int nxt = (p + 1 < ol) ? gclampi(tg[(int64_t) (p + 1) * P.gs1], 0, N - 1) : 0;
        const R h2 = act[k] ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : R(0);
        const R dp = (act[k] && p >= 1) ? fmax(tr[(int64_t) cur * P.ts0 + (int64_t) prv * P.ts1] * L2E, LZ) : LZ;
        const R dn = (p + 1 < ol) ? fmax(tr[(int64_t) nxt * P.ts0 + (int64_t) cur * P.ts1] * L2E, LZ) : LZ;
        H2[k] = (double) h2;
        Dx[k] = (double) (beta ? dn : dp);
        in[k] = (const R *) P.inputs + (int64_t) b * P.is1 + (int64_t) cur * P.is2;
        if (STORE && !beta && p < S) {
            V2<R> u = {h2, dp};
            reinterpret_cast<V2<R> *>(W.asu)[(int64_t) b * S + p] = u;
            int2 ii = {cur, prv};
            reinterpret_cast<int2 *>(W.asi)[(int64_t) b * S + p] = ii;
        }
    }
    R *score_out = (R *) (beta ? O.aligned_scores : O.aligned_scores_alpha);
    if (len < 1 || ol < 1) {
        if (tid == 0 && score_out) score_out[b] = Num<R>::ninf();
        return;
    }
    const double kZ = -1e30, L2Ed = 1.4426950408889634;
    auto lse2d = [&](double x, double y) {
        const double m = fmax(x, y);
        const R d = (R) (fmin(x, y) - m);
        return m + (double) Num<R>::log2(R(1) + Num<R>::exp2(d));
    };
    // renormalise now and then: the log domain is offset free (the stored states are doubles: AlignedState)
    auto renorm = [&](double (&v)[KP], double &C) {
        R m = LZ;
#pragma unroll
        for (int k = 0; k < KP; ++k) m = fmax(m, (R) v[k]);
        m = wave_allmax(m);
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        R mm = red[0];
        for (int w = 1; w < 16; ++w) mm = fmax(mm, red[w]);
        if (mm > R(-1e29)) {
#pragma unroll
            for (int k = 0; k < KP; ++k) v[k] = fmax(v[k] - (double) mm, kZ);
            C += (double) mm;
        }
        __syncthreads();
    };
    double C = 0.0, v[KP];
    if (!beta) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int p = tid + 1024 * k;
            v[k] = (p == 0 && act[k]) ? fmax((double) in[k][0] * L2Ed, kZ) : kZ;
            if (STORE && p < S) out[p] = (AlignedState) v[k];
        }
        for (int t = 1; t < len; ++t) {
            double *rw = (t & 1) ? row1 : row0;
#pragma unroll
            for (int k = 0; k < KP; ++k) { const int p = tid + 1024 * k; if (p < S) rw[p + 1] = v[k]; }
            if (tid == 0) rw[0] = kZ;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int p = tid + 1024 * k;
                const double em = act[k] ? (double) in[k][(int64_t) t * P.is0] * L2Ed : kZ;
                const double left = p < S ? rw[p] : kZ;
                v[k] = fmax(em + lse2d(v[k] + H2[k], left + Dx[k]), kZ);
            }
            if ((t & 15) == 0) renorm(v, C);
            if (STORE) {
#pragma unroll
                for (int k = 0; k < KP; ++k) { const int p = tid + 1024 * k; if (p < S) out[(int64_t) t * S + p] = (AlignedState) v[k]; }
            }
        }
        if (score_out) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KP; ++k) { const int p = tid + 1024 * k; if (p < S) row0[p] = v[k]; }
            __syncthreads();
            if (tid == 0) {
                const double sc = C + row0[ol - 1];
                score_out[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int p = tid + 1024 * k;
            v[k] = (p == ol - 1) ? 0.0 : kZ;
            if (STORE && p < S) out[(int64_t) (len - 1) * S + p] = (AlignedState) v[k];
        }
        for (int t = len - 1; t >= 1; --t) {
            double *rw = (t & 1) ? row1 : row0;
            double y[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int p = tid + 1024 * k;
                const double em = act[k] ? (double) in[k][(int64_t) t * P.is0] * L2Ed : kZ;
                y[k] = fmax(em + v[k], kZ);
                if (p < S) rw[p] = y[k];
            }
            if (tid == 0) rw[S] = kZ;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int p = tid + 1024 * k;
                const double right = p < S ? rw[p + 1] : kZ;
                v[k] = fmax(lse2d(y[k] + H2[k], right + Dx[k]), kZ);
            }
            if ((t & 15) == 0) renorm(v, C);
            if (STORE) {
#pragma unroll
                for (int k = 0; k < KP; ++k) { const int p = tid + 1024 * k; if (p < S) out[(int64_t) (t - 1) * S + p] = (AlignedState) v[k]; }
            }
        }
        if (score_out && tid == 0) {
            const double em = act[0] ? (double) in[0][0] * L2Ed : kZ;
            const double sc = C + (em + v[0]);
            score_out[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
        }
    }
}

// Gradient of the same: grid = (B, nchunks), block = 256.  The workgroup walks the frames of its chunk ONE AT A TIME, thread tid owns
// positions tid, tid + 256, ... (KQ = 16 of them: S <= 4096; 32: S <= 8192) with their edge-posterior sums in registers; per frame two
// block reductions (maximum, sum: the reference's masked softmax over positions), the posteriors scattered to the labels
// through ONE fixed-point LDS row of N words (dynamic LDS; integer adds commute: deterministic, repeated labels included), read back and
// added to grad_inputs.  Edge posteriors per (b, chunk) go to gHD as from bwd_aligned_kernel (aligned_tr_scatter_fx_kernel follows, or
// beyond 2048 labels the hash-table scatter aligned_tr_scatter_kernel).  Restates force_aligned_lattice.cpp:156-264.
extern __shared__ __attribute__((aligned(16))) unsigned char strip_row_bytes[];
template <typename R, int KQ>
__global__ void __launch_bounds__(256) bwd_aligned_strip_kernel(Problem P, State W, BwdArgs A, R *gHD, int add_to_inputs) {
    typedef typename FrameFix<R>::T FX;
    FX *fxl = reinterpret_cast<FX *>(strip_row_bytes);
    __shared__ R red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int S = P.S, T = P.T, N = P.N;
    const R LZ = Num<R>::logzero();
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const R g0 = (R) ((double) ((const R *) (A.grad_aligned ? A.grad_aligned : A.grad_full))[(int64_t) b * A.gstride] * A.gscale);
    const R ga = (A.grad_aligned || !A.neg_aligned) ? g0 : -g0;
    const int2 *asi = reinterpret_cast<const int2 *>(W.asi) + (int64_t) b * S;
    const V2<R> *asu = reinterpret_cast<const V2<R> *>(W.asu) + (int64_t) b * S;
    for (int q = tid; q < N; q += 256) fxl[q] = 0;
    R H2[KQ], Dp[KQ], accH[KQ], accD[KQ];
    int tgt[KQ];
    bool act[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int p = tid + 256 * k;
        act[k] = p < ol;
        const V2<R> u = p < S ? asu[p] : V2<R>{0, LZ};
        H2[k] = u.x; Dp[k] = u.y;
        tgt[k] = act[k] ? asi[p].x : 0;
        accH[k] = 0; accD[k] = 0;
    }
    __syncthreads();
    const AlignedState *abp = (const AlignedState *) W.ab + (int64_t) b * T * S;
    const AlignedState *bbp = (const AlignedState *) W.bb + (int64_t) b * T * S;
    auto block_max = [&](R v) -> R {
        v = wave_allmax(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        const R r = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        __syncthreads();
        return r;
    };
    auto block_sum = [&](R v) -> R {
        v = wave_allsum(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        const R r = (red[0] + red[1]) + (red[2] + red[3]);      // fixed order
        __syncthreads();
        return r;
    };
    const int t0 = chunk * A.chunk, t1 = min(min(T, t0 + A.chunk), len);
    for (int t = t0; t < t1; ++t) {
        double gs[KQ];
        R m = LZ + LZ;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int p = tid + 256 * k;
            gs[k] = p < S ? abp[(int64_t) t * S + p] + bbp[(int64_t) t * S + p] : -2e30;
            m = fmax(m, (R) gs[k]);
        }
        m = block_max(m);
        R e[KQ], z = 0;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            e[k] = (m > R(-1e29) && act[k]) ? Num<R>::exp2((R) (gs[k] - (double) m)) : R(0);
            z += e[k];
        }
        z = block_sum(z);
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            const int p = tid + 256 * k;
            const R post = (z > 0 && act[k]) ? e[k] / z : R(0);
            if (post != R(0)) atomicAdd(&fxl[tgt[k]], FrameFix<R>::to(post));
            if (t >= 1 && act[k]) {
                // shares of the two incoming edges from their difference (formed in double): bwd_aligned_long_kernel
                const double ap = abp[(int64_t) (t - 1) * S + p];
                const double al = p >= 1 ? abp[(int64_t) (t - 1) * S + p - 1] : 0.0;
                const R d = (R) ((al + (double) Dp[k]) - (ap + (double) H2[k]));
                const R tt = Num<R>::exp2(-fabs(d));
                const R big = R(1) / (R(1) + tt), small = tt * big;
                accH[k] += post * (d <= R(0) ? big : small);
                accD[k] += post * (d <= R(0) ? small : big);
            }
        }
        __syncthreads();
        for (int lab = tid; lab < N; lab += 256) {
            const FX fv = fxl[lab];
            if (fv != 0) {
                fxl[lab] = 0;
                R *gin = (R *) A.grad_inputs + ((int64_t) t * P.B + b) * N + lab;
                const R add = ga * FrameFix<R>::from(fv);
                *gin = add_to_inputs ? *gin + add : add;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int p = tid + 256 * k;
        if (p < S) {
            R *dst = gHD + ((int64_t) b * A.nchunks + chunk) * 2 * S;
            dst[p] = accH[k];
            dst[S + p] = accD[k];
        }
    }
}

// scale of the 64-bit fixed-point accumulators in memory (sums over the whole batch): 2^36 (fp32) / 2^40 (fp64)
template <typename R> struct GlobalFix { static constexpr double scale = sizeof(R) == 4 ? 68719476736.0 : 1099511627776.0; };   // 2^36 / 2^40

// ------------------------------------------------------------------ gradient: aligned lattice, small alphabet + long targets
// N <= 64, 64 < S <= 512 (the lattices of letter-based models: a few dozen labels, targets of hundreds of positions).
// grid = (B, nchunks), block = 256.  Wave w handles frames t0+w, t0+w+4, ...; lane l owns the K consecutive positions
// K l .. K l + K - 1 (as aligned_long_kernel).  The aligned posteriors of a frame are scattered to the N labels with
// fixed-point LDS adds (integer adds commute: repeated labels give bit-identical sums, no O(S^2) de-duplication as in
// bwd_aligned_kernel below, which has to serve N = 10^4) and added to the frame's grad_inputs row; the stay / arrive
// edge posteriors go the same way into ONE [N][N] fixed-point tile per workgroup, written out as a float tile that
// add_tiles_kernel sums over (b, chunk) in a fixed order -- no single-workgroup scatter over the whole batch
// (aligned_tr_scatter_kernel: 2.5 ms at T = 1000 B = 64 S = 200).  Restates force_aligned_lattice.cpp:156-264.
// NL = 256 (64 < N <= 256): label rows of 256 words; the edge posteriors go straight into the [N][N] 64-bit fixed-point
// accumulator in memory (`gfx`, as aligned_tr_scatter_fx_kernel) instead of an LDS tile.
// SR: type of the stored aligned states (AlignedState = double when the long-target kernels wrote them, S > 64; the
// problem's type when the one-wavefront chains of the small path did, S <= 64).  With SR = double and R = float the sums
// ab + bb and the differences between neighbouring states are formed in double and only then rounded.
template <typename R, int K, int NL, typename SR>
__global__ void __launch_bounds__(256) bwd_aligned_long_kernel(Problem P, State W, BwdArgs A, R *tiles, int add_to_inputs,
                                                               unsigned long long *gfx) {
    typedef typename FrameFix<R>::T FX;
    __shared__ FX fxI[4][NL];
    __shared__ unsigned long long fxT[NL == 64 ? 64 * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int S = P.S, T = P.T, N = P.N;
    const R LZ = Num<R>::logzero();
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const R g0 = (R) ((double) ((const R *) (A.grad_aligned ? A.grad_aligned : A.grad_full))[(int64_t) b * A.gstride] * A.gscale);
    const R ga = (A.grad_aligned || !A.neg_aligned) ? g0 : -g0;
    const int2 *asi = reinterpret_cast<const int2 *>(W.asi) + (int64_t) b * S;
    const V2<R> *asu = reinterpret_cast<const V2<R> *>(W.asu) + (int64_t) b * S;
    if (NL == 64) for (int q = threadIdx.x; q < N * N; q += 256) fxT[q] = 0;
    for (int q = lane; q < NL; q += 64) fxI[wave][q] = 0;
    R H2[K], Dp[K];
    double accH[K], accD[K];
    int tgt[K], prv[K];
    bool act[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = lane * K + k;
        act[k] = p < ol;
        const V2<R> u = p < S ? asu[p] : V2<R>{0, LZ};
        const int2 ii = p < S ? asi[p] : int2{0, 0};
        H2[k] = u.x; Dp[k] = u.y;
        tgt[k] = act[k] ? ii.x : lane;           // (positions past the target add 0: each lane to a word of its own)
        prv[k] = ii.y;
        accH[k] = 0; accD[k] = 0;
    }
    __syncthreads();
    const SR *abp = (const SR *) W.ab + (int64_t) b * T * S;
    const SR *bbp = (const SR *) W.bb + (int64_t) b * T * S;
    const SR LZs = (SR) LZ;
    const int t0 = chunk * A.chunk, t1 = min(min(T, t0 + A.chunk), len);
    for (int t = t0 + wave; t < t1; t += 4) {
        SR gs[K];
        R gam[K], m = LZ;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int p = lane * K + k;
            gs[k] = p < S ? abp[(int64_t) t * S + p] + bbp[(int64_t) t * S + p] : LZs + LZs;
            m = fmax(m, (R) gs[k]);
        }
        m = wave_allmax(m);                      // (any value near the largest sum serves as the common reference)
        R z = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            gam[k] = (m > R(-1e29)) ? Num<R>::exp2((R) (gs[k] - (SR) m)) : R(0);
            z += gam[k];
        }
        z = wave_allsum(z);
        SR apl = LZs;                            // alpha-bar of the previous frame at the position left of this lane's first
        SR ap[K];
        if (t >= 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) ap[k] = act[k] ? abp[(int64_t) (t - 1) * S + lane * K + k] : LZs;
            apl = prev_lane_or_zero<SR>(ap[K - 1]);
            if (lane == 0) apl = SR(0);          // (position 0 has no arrive edge: Dp is log-zero there)
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const R post = (z > 0 && act[k]) ? gam[k] / z : R(0);
            atomicAdd(&fxI[wave][tgt[k]], FrameFix<R>::to(post));
            if (t >= 1 && act[k]) {
                // stay / arrive shares of the state posterior: softmax over the two incoming edges, from their DIFFERENCE
                // (formed in the stored type): 1 / (1 + 2^-|d|) and its complement
                const SR al = k == 0 ? apl : ap[k - 1];
                const R d = (R) ((al + (SR) Dp[k]) - (ap[k] + (SR) H2[k]));
                const R tt = Num<R>::exp2(-fabs(d));
                const R big = R(1) / (R(1) + tt), small = tt * big;
                accH[k] += (double) (post * (d <= R(0) ? big : small));
                accD[k] += (double) (post * (d <= R(0) ? small : big));
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NL / 64; ++q) {
            const int lab = lane + 64 * q;
            const FX fv = fxI[wave][lab];
            fxI[wave][lab] = 0;
            if (lab < N && (NL == 64 || fv != 0)) {
                R *gin = (R *) A.grad_inputs + ((int64_t) t * P.B + b) * N + lab;
                const R add = ga * FrameFix<R>::from(fv);
                *gin = add_to_inputs ? *gin + add : add;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // this workgroup's edge posteriors -> one [N][N] tile: stay (O_s, O_s), arrive (O_s, O_{s-1})   (force_aligned_lattice.cpp:204-231)
    if constexpr (NL == 64) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (act[k]) {
                if (accH[k] != 0.0) atomicAdd(&fxT[tgt[k] * N + tgt[k]], (unsigned long long) __double2ll_rn(accH[k] * Num<R>::kFix));
                if (lane * K + k >= 1 && accD[k] != 0.0)
                    atomicAdd(&fxT[tgt[k] * N + prv[k]], (unsigned long long) __double2ll_rn(accD[k] * Num<R>::kFix));
            }
        }
        __syncthreads();
        R *tile = tiles + ((int64_t) b * A.nchunks + chunk) * N * N;
        for (int q = threadIdx.x; q < N * N; q += 256)
            tile[q] = (R) ((double) ga * ((double) (long long) fxT[q] * (1.0 / Num<R>::kFix)));
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (act[k]) {
                const long long qh = __double2ll_rn((double) ga * accH[k] * GlobalFix<R>::scale);
                const long long qd = __double2ll_rn((double) ga * accD[k] * GlobalFix<R>::scale);
                if (qh != 0) atomicAdd(&gfx[(int64_t) tgt[k] * N + tgt[k]], (unsigned long long) qh);
                if (lane * K + k >= 1 && qd != 0) atomicAdd(&gfx[(int64_t) tgt[k] * N + prv[k]], (unsigned long long) qd);
            }
        }
    }
}

// out[k] (+)= sum over the G tiles in a fixed order (deterministic).  grid = ceil(n / 32), block = 1024 = 32 elements x
// 32 tile groups: thread (e, grp) sums tiles grp, grp + 32, ... (8 loads in flight), then a fixed-order combine in LDS.
template <typename R>
__global__ void __launch_bounds__(1024) add_tiles_kernel(const R *tiles, int G, int n, R *out, int accumulate) {
    __shared__ R part[32][33];
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int k = min((int) blockIdx.x * 32 + e, n - 1);
    R a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0;
    for (int g = grp; g < G; g += 32 * 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int gg = g + 32 * q;
            const R v = tiles[(int64_t) min(gg, G - 1) * n + k];
            a[q] += gg < G ? v : R(0);
        }
    }
    part[grp][e] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (grp == 0 && (int) blockIdx.x * 32 + e < n) {
        R t = part[0][e];
#pragma unroll
        for (int q = 1; q < 32; ++q) t += part[q][e];
        out[k] = accumulate ? out[k] + t : t;
    }
}

// ------------------------------------------------------------------ gradient: aligned lattice (any N, S <= 1024)
// grid = (B, nchunks), block = 256.  Wave w handles frames t0+w, t0+w+4, ...; lane l covers target positions
// l, l+64, ...  Duplicate labels inside an utterance are folded onto their FIRST occurrence in a fixed order,
// so the read-modify-write of grad_inputs needs no atomics and is deterministic.
// Edge posteriors are written per (b, chunk) to gHD[(b*nchunks+chunk)][2][S].
template <typename R, typename SR>
__global__ void __launch_bounds__(256) bwd_aligned_kernel(Problem P, State W, BwdArgs A, R *gHD, int add_to_inputs) {
    constexpr int MAXK = 16;                       // S <= 1024
    __shared__ R post_s[4][1024];
    __shared__ int first_s[1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int S = P.S, T = P.T, N = P.N;
    const R LZ = Num<R>::logzero();
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const R g0 = (R) ((double) ((const R *) (A.grad_aligned ? A.grad_aligned : A.grad_full))[(int64_t) b * A.gstride] * A.gscale);
    const R ga = (A.grad_aligned || !A.neg_aligned) ? g0 : -g0;
    const int K = (S + 63) / 64;
    const int2 *asi = reinterpret_cast<const int2 *>(W.asi) + (int64_t) b * S;
    const V2<R> *asu = reinterpret_cast<const V2<R> *>(W.asu) + (int64_t) b * S;
    // first occurrence of each position's label (fixed order -> deterministic)
    for (int s = threadIdx.x; s < S; s += 256) {
        int f = s;
        if (s < ol) {
            const int lab = asi[s].x;
            for (int q = 0; q < s; ++q) if (asi[q].x == lab) { f = q; break; }
        }
        first_s[s] = f;
    }
    __syncthreads();
    R H2[MAXK], Dp[MAXK], accH[MAXK], accD[MAXK];
    int tgt[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        int s = lane + 64 * k;
        bool v = k < K && s < S;
        V2<R> u = v ? asu[s] : V2<R>{0, LZ};
        H2[k] = u.x; Dp[k] = u.y;
        tgt[k] = v ? asi[s].x : 0;
        accH[k] = 0; accD[k] = 0;
    }
    const SR *abp = (const SR *) W.ab + (int64_t) b * T * S;
    const SR *bbp = (const SR *) W.bb + (int64_t) b * T * S;
    const SR LZs = (SR) LZ;
    const int t0 = chunk * A.chunk, t1 = min(T, t0 + A.chunk);
    for (int t = t0 + wave; t < t1; t += 4) {
        if (t >= len) continue;
        SR gs[MAXK];
        R gam[MAXK], m = LZ;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            int s = lane + 64 * k;
            gs[k] = (k < K && s < S) ? abp[(int64_t) t * S + s] + bbp[(int64_t) t * S + s] : LZs + LZs;
            m = fmax(m, (R) gs[k]);
        }
        m = wave_allmax(m);
        R z = 0;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            gam[k] = (k < K && m > R(-1e29)) ? Num<R>::exp2((R) (gs[k] - (SR) m)) : R(0);
            z += gam[k];
        }
        z = wave_allsum(z);
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            int s = lane + 64 * k;
            R post = (z > 0 && k < K && s < ol) ? gam[k] / z : R(0);
            if (k < K && s < S) post_s[wave][s] = post;
            if (t >= 1 && k < K && s < ol) {
                // (shares of the two incoming edges from their difference, formed in the stored type: bwd_aligned_long_kernel)
                const SR ap = abp[(int64_t) (t - 1) * S + s];
                const SR al = s >= 1 ? abp[(int64_t) (t - 1) * S + s - 1] : SR(0);
                const R d = (R) ((al + (SR) Dp[k]) - (ap + (SR) H2[k]));
                const R tt = Num<R>::exp2(-fabs(d));
                const R big = R(1) / (R(1) + tt), small = tt * big;
                accH[k] += post * (d <= R(0) ? big : small);
                accD[k] += post * (d <= R(0) ? small : big);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // owner (first occurrence) sums its duplicates in ascending order and updates grad_inputs[t][b][label]
        R *gin = (R *) A.grad_inputs + ((int64_t) t * P.B + b) * N;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            int s = lane + 64 * k;
            if (k < K && s < ol && first_s[s] == s) {
                R sum = post_s[wave][s];
                for (int q = s + 1; q < ol; ++q) if (first_s[q] == s) sum += post_s[wave][q];
                R add = ga * sum;
                gin[tgt[k]] = add_to_inputs ? gin[tgt[k]] + add : add;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // reduce the 4 waves' edge posteriors and write this chunk's slice
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            int s = lane + 64 * k;
            if (k < K && s < S) post_s[wave][s] = pass == 0 ? accH[k] : accD[k];
        }
        __syncthreads();
        R *dst = gHD + (((int64_t) b * A.nchunks + chunk) * 2 + pass) * S;
        for (int s = threadIdx.x; s < S; s += 256)
            dst[s] = (post_s[0][s] + post_s[1][s]) + (post_s[2][s] + post_s[3][s]);
        __syncthreads();
    }
}

// Medium alphabets (64 < N <= 2048): the same scatter with one workgroup PER UTTERANCE and 64-bit fixed-point atomic adds
// into an [N][N] accumulator in memory (integer adds commute: deterministic; the float result is formed once, by
// fx_to_grad_kernel) -- the single workgroup below walks the batch utterance by utterance (713 us at B = 64, S = 30).
template <typename R>
__global__ void __launch_bounds__(256) aligned_tr_scatter_fx_kernel(Problem P, State W, BwdArgs A, const R *gHD, unsigned long long *fx) {
    const int S = P.S, N = P.N, b = blockIdx.x;
    const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
    const R g0 = (R) ((double) ((const R *) (A.grad_aligned ? A.grad_aligned : A.grad_full))[(int64_t) b * A.gstride] * A.gscale);
    const R ga = (A.grad_aligned || !A.neg_aligned) ? g0 : -g0;
    const int2 *asi = reinterpret_cast<const int2 *>(W.asi) + (int64_t) b * S;
    for (int e = threadIdx.x; e < 2 * S; e += 256) {
        const int pass = e / S, s = e - pass * S;
        const bool valid = pass == 0 ? (s < ol) : (s >= 1 && s < ol);
        if (!valid) continue;
        R v = 0;
        for (int c = 0; c < A.nchunks; ++c) v += gHD[(((int64_t) b * A.nchunks + c) * 2 + pass) * S + s];
        const int2 ii = asi[s];
        const long long q = __double2ll_rn((double) ga * (double) v * GlobalFix<R>::scale);
        if (q != 0) atomicAdd(&fx[(int64_t) ii.x * N + (pass == 0 ? ii.x : ii.y)], (unsigned long long) q);
    }
}
template <typename R>
__global__ void __launch_bounds__(256) fx_to_grad_kernel(const unsigned long long *fx, int64_t n, R *out, int accumulate) {
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const R v = (R) ((double) (long long) fx[k] * (1.0 / GlobalFix<R>::scale));
    out[k] = accumulate ? out[k] + v : v;
}

// scatter the aligned edge posteriors into grad_transition where an N x N fixed-point image is too large (N > 2048): ONE workgroup, the
// batch's B x 2 S entries a thousand at a time through a hash table in LDS -- 64-bit keys claimed by compare-and-swap, values added as
// 64-bit fixed point (integer sums: exact, so the order in which the threads arrive does not matter) -- then every occupied slot adds its
// sum to its element of the gradient: one thread per key and round, rounds in order -> deterministic.
// (Before: utterance by utterance with an O(S^2) search for duplicates, 64 rounds of dependent memory latency at B = 64: 0.72 ms at
// N = 3000, S = 30; searching 34 utterances' entries at once was no faster -- 2 000 dependent LDS reads per thread: 0.84 ms.)
// keys: stay  (O_s, O_s)      <- gH[s]   for s < ol
//       enter (O_s, O_{s-1})  <- gD[s]   for 1 <= s < ol
template <typename R>
__global__ void __launch_bounds__(1024) aligned_tr_scatter_kernel(Problem P, State W, BwdArgs A, const R *gHD, R *out, int accumulate) {
    constexpr int HT = 2048;                        // slots: twice the entries of a round
    __shared__ unsigned long long key_s[HT];        // 0 = free; key + 1 otherwise
    __shared__ unsigned long long val_s[HT];
    const int S = P.S, N = P.N, B = P.B;
    if (!accumulate) {
        for (int64_t k = threadIdx.x; k < (int64_t) N * N; k += blockDim.x) out[k] = 0;
        __syncthreads();
    }
    const int64_t total = (int64_t) B * 2 * S;
    for (int64_t e0 = 0; e0 < total; e0 += 1024) {
        for (int h = threadIdx.x; h < HT; h += 1024) { key_s[h] = 0ull; val_s[h] = 0ull; }
        __syncthreads();
        const int64_t e = e0 + threadIdx.x;
        if (e < total) {
            const int b = (int) (e / (2 * S)), idx = (int) (e - (int64_t) b * 2 * S), pass = idx / S, s = idx - pass * S;
            const int ol = P.tg_len ? gclampi(P.tg_len[b], 0, S) : S;
            const bool valid = pass == 0 ? (s < ol) : (s >= 1 && s < ol);
            if (valid) {
                const R g0 = (R) ((double) ((const R *) (A.grad_aligned ? A.grad_aligned : A.grad_full))[(int64_t) b * A.gstride] * A.gscale);
                const R ga = (A.grad_aligned || !A.neg_aligned) ? g0 : -g0;
                R v = 0;
                for (int c = 0; c < A.nchunks; ++c) v += gHD[(((int64_t) b * A.nchunks + c) * 2 + pass) * S + s];
                const int2 ii = (reinterpret_cast<const int2 *>(W.asi) + (int64_t) b * S)[s];
                const long long q = __double2ll_rn((double) ga * (double) v * GlobalFix<R>::scale);
                if (q != 0) {
                    const unsigned long long key = (unsigned long long) ((long long) ii.x * N + (pass == 0 ? ii.x : ii.y)) + 1ull;
                    unsigned h = (unsigned) ((key * 0x9E3779B97F4A7C15ull) >> 53) & (HT - 1);
                    for (;;) {      // (at most 1024 keys in 2048 slots: always ends)
                        const unsigned long long prev = atomicCAS(&key_s[h], 0ull, key);
                        if (prev == 0ull || prev == key) { atomicAdd(&val_s[h], (unsigned long long) q); break; }
                        h = (h + 1) & (HT - 1);
                    }
                }
            }
        }
        __syncthreads();
        for (int h = threadIdx.x; h < HT; h += 1024)
            if (key_s[h] != 0ull) out[key_s[h] - 1ull] += (R) ((double) (long long) val_s[h] * (1.0 / GlobalFix<R>::scale));
        __syncthreads();
    }
}

}  // namespace

template <typename R>
hipError_t launch_fwd_aligned_generic(const Problem &P, const State &W, const FwdOut &O, int ali_mask, bool store, hipStream_t stream) {
    if (ali_mask && P.S > 1024) {
        // very long targets (up to 8192 positions): four / eight positions per thread, the frame's states through LDS (two rows of
        // S + 2 doubles: 131 KB at S = 8192 -- what a compute unit's LDS holds is what bounds the target length here)
        if (P.S > kMaxTargets) return hipErrorInvalidValue;
        dim3 grid(P.B, __builtin_popcount(ali_mask));
        const size_t dyn = (size_t) 2 * (P.S + 2) * sizeof(double);
        if (P.S <= 4096) {
            (void) hipFuncSetAttribute((const void *) aligned_strip_kernel<R, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn);
            (void) hipFuncSetAttribute((const void *) aligned_strip_kernel<R, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn);
            if (store) hipLaunchKernelGGL((aligned_strip_kernel<R, true, 4>), grid, dim3(1024), dyn, stream, P, W, O, ali_mask);
            else hipLaunchKernelGGL((aligned_strip_kernel<R, false, 4>), grid, dim3(1024), dyn, stream, P, W, O, ali_mask);
        } else {
            (void) hipFuncSetAttribute((const void *) aligned_strip_kernel<R, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn);
            (void) hipFuncSetAttribute((const void *) aligned_strip_kernel<R, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn);
            if (store) hipLaunchKernelGGL((aligned_strip_kernel<R, true, 8>), grid, dim3(1024), dyn, stream, P, W, O, ali_mask);
            else hipLaunchKernelGGL((aligned_strip_kernel<R, false, 8>), grid, dim3(1024), dyn, stream, P, W, O, ali_mask);
        }
    } else if (ali_mask) {
        const int threads = ((P.S + 63) / 64) * 64;
        if (threads > 1024) return hipErrorInvalidValue;
        dim3 grid(P.B, __builtin_popcount(ali_mask));
        // up to 512 target positions: one wavefront per chain, 2 / 4 / 8 positions per lane; beyond: one position per
        // thread and a workgroup barrier per frame
        const double fr = (double) (P.T - 1) * (double) P.is0 * sizeof(R), ln = (double) (P.N - 1) * (double) P.is2 * sizeof(R);
        const bool off32 = P.is0 >= 0 && P.is2 >= 0 && fr < 4294967296.0 && ln < 2147483648.0 &&
                           (double) P.T * P.S * sizeof(AlignedState) < 4294967296.0;
        const char ak = knobs().aligned_kernel;        // developer A/B (ASG_ALIGNED_KERNEL): "long" (K positions per lane), "wide" (barrier per frame), "pipe"
        const bool use_pipe = off32 && ((P.S > 256 && !(ak == 'l' || ak == 'w')) || ak == 'p');
        if (use_pipe) {
            if (store) hipLaunchKernelGGL((aligned_pipe_kernel<R, true>), grid, dim3(threads), 0, stream, P, W, O, ali_mask);
            else hipLaunchKernelGGL((aligned_pipe_kernel<R, false>), grid, dim3(threads), 0, stream, P, W, O, ali_mask);
        } else if (!off32 || P.S > 512 || ak == 'w') {
            if (store) hipLaunchKernelGGL((aligned_wide_kernel<R, true>), grid, dim3(threads), 0, stream, P, W, O, ali_mask);
            else hipLaunchKernelGGL((aligned_wide_kernel<R, false>), grid, dim3(threads), 0, stream, P, W, O, ali_mask);
        } else if (P.S <= 128) {
            if (store) hipLaunchKernelGGL((aligned_long_kernel<R, 2, true>), grid, dim3(64), 0, stream, P, W, O, ali_mask);
            else hipLaunchKernelGGL((aligned_long_kernel<R, 2, false>), grid, dim3(64), 0, stream, P, W, O, ali_mask);
        } else if (P.S <= 256) {
            if (store) hipLaunchKernelGGL((aligned_long_kernel<R, 4, true>), grid, dim3(64), 0, stream, P, W, O, ali_mask);
            else hipLaunchKernelGGL((aligned_long_kernel<R, 4, false>), grid, dim3(64), 0, stream, P, W, O, ali_mask);
        } else {
            if (store) hipLaunchKernelGGL((aligned_long_kernel<R, 8, true>), grid, dim3(64), 0, stream, P, W, O, ali_mask);
            else hipLaunchKernelGGL((aligned_long_kernel<R, 8, false>), grid, dim3(64), 0, stream, P, W, O, ali_mask);
        }
    }
    return hipGetLastError();
}

template <typename R>
hipError_t launch_bwd_aligned_generic(const Problem &P, const State &W, const BwdArgs &A, const GenericBwdLayout &Y, bool have_full,
                                      bool fx_cleared, hipStream_t stream) {
    const size_t e = sizeof(R);
    R *gHD = (R *) Y.gHD, *atiles = (R *) Y.atiles, *gtr = (R *) A.grad_transition;
    const bool do_ali = true;
    if (do_ali) {
        if (P.S > kMaxTargets) return hipErrorInvalidValue;
        if (!have_full) (void) zero_async(A.grad_inputs, (size_t) P.T * P.B * P.N * e, stream);
        unsigned long long *nofx = nullptr;
        if (P.S > 1024) {
            // very long targets: frame-by-frame workgroups, label scatter through a fixed-point LDS row of N words; the edge posteriors
            // into the 64-bit fixed-point accumulator, as for the medium alphabets (N <= 2048), or through the hash-table scatter
            const size_t row = ((size_t) P.N * sizeof(typename FrameFix<R>::T) + 255) & ~(size_t) 255;
            if (row > 150 * 1024) return hipErrorInvalidValue;          // (N beyond ~19 000 / 38 000 labels with targets beyond 1024 positions)
            if (P.S <= 4096) {
                (void) hipFuncSetAttribute((const void *) bwd_aligned_strip_kernel<R, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) row);
                hipLaunchKernelGGL((bwd_aligned_strip_kernel<R, 16>), dim3(P.B, A.nchunks), dim3(256), row, stream, P, W, A, gHD, 1);
            } else {
                (void) hipFuncSetAttribute((const void *) bwd_aligned_strip_kernel<R, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) row);
                hipLaunchKernelGGL((bwd_aligned_strip_kernel<R, 32>), dim3(P.B, A.nchunks), dim3(256), row, stream, P, W, A, gHD, 1);
            }
            if (P.N <= 2048) {
                unsigned long long *fx = (unsigned long long *) atiles;
                const int64_t n2 = (int64_t) P.N * P.N;
                if (!fx_cleared) (void) zero_async(fx, (size_t) n2 * 8, stream);
                hipLaunchKernelGGL((aligned_tr_scatter_fx_kernel<R>), dim3(P.B), dim3(256), 0, stream, P, W, A, (const R *) gHD, fx);
                hipLaunchKernelGGL((fx_to_grad_kernel<R>), dim3((unsigned) ((n2 + 255) / 256)), dim3(256), 0, stream,
                                   (const unsigned long long *) fx, n2, gtr, have_full ? 1 : 0);
            } else {
                // (without a full-lattice part before it the gradient starts from zero: a memset, not one workgroup walking N x N elements)
                if (!have_full) (void) zero_async(gtr, (size_t) P.N * P.N * e, stream);
                hipLaunchKernelGGL((aligned_tr_scatter_kernel<R>), dim3(1), dim3(1024), 0, stream, P, W, A, gHD, gtr, 1);
            }
        } else if (P.N <= 64 && P.S > 64 && P.S <= 1024) {
            dim3 grid(P.B, A.nchunks);
            if (P.S <= 128) hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 2, 64, AlignedState>), grid, dim3(256), 0, stream, P, W, A, atiles, 1, nofx);
            else if (P.S <= 256) hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 4, 64, AlignedState>), grid, dim3(256), 0, stream, P, W, A, atiles, 1, nofx);
            else if (P.S <= 512) hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 8, 64, AlignedState>), grid, dim3(256), 0, stream, P, W, A, atiles, 1, nofx);
            else hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 16, 64, AlignedState>), grid, dim3(256), 0, stream, P, W, A, atiles, 1, nofx);
            const int n2 = P.N * P.N;
            hipLaunchKernelGGL((add_tiles_kernel<R>), dim3((n2 + 31) / 32), dim3(1024), 0, stream, (const R *) atiles,
                               P.B * A.nchunks, n2, gtr, have_full ? 1 : 0);
        } else if (P.N > 64 && P.N <= 256 && P.S <= 1024) {
            // medium alphabet, targets of any length (short ones included: the de-duplicating kernel below + its scatter took
            // 95 us at T=400 B=64 N=128 S=30, this 35)
            dim3 grid(P.B, A.nchunks);
            unsigned long long *fx = (unsigned long long *) atiles;
            const int64_t n2 = (int64_t) P.N * P.N;
            if (!fx_cleared) (void) zero_async(fx, (size_t) n2 * 8, stream);
            // (S <= 64: the states are the one-wavefront chains' (asg_chains.h), stored in the problem's type)
            if (P.S <= 64) hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 2, 256, R>), grid, dim3(256), 0, stream, P, W, A, (R *) nullptr, 1, fx);
            else if (P.S <= 128) hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 2, 256, AlignedState>), grid, dim3(256), 0, stream, P, W, A, (R *) nullptr, 1, fx);
            else if (P.S <= 256) hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 4, 256, AlignedState>), grid, dim3(256), 0, stream, P, W, A, (R *) nullptr, 1, fx);
            else if (P.S <= 512) hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 8, 256, AlignedState>), grid, dim3(256), 0, stream, P, W, A, (R *) nullptr, 1, fx);
            else hipLaunchKernelGGL((bwd_aligned_long_kernel<R, 16, 256, AlignedState>), grid, dim3(256), 0, stream, P, W, A, (R *) nullptr, 1, fx);
            hipLaunchKernelGGL((fx_to_grad_kernel<R>), dim3((unsigned) ((n2 + 255) / 256)), dim3(256), 0, stream,
                               (const unsigned long long *) fx, n2, gtr, have_full ? 1 : 0);
        } else {
            if (P.S <= 64) hipLaunchKernelGGL((bwd_aligned_kernel<R, R>), dim3(P.B, A.nchunks), dim3(256), 0, stream, P, W, A, gHD, 1);
            else hipLaunchKernelGGL((bwd_aligned_kernel<R, AlignedState>), dim3(P.B, A.nchunks), dim3(256), 0, stream, P, W, A, gHD, 1);
            if (P.N > 64 && P.N <= 2048) {
                unsigned long long *fx = (unsigned long long *) atiles;
                const int64_t n2 = (int64_t) P.N * P.N;
                if (!fx_cleared) (void) zero_async(fx, (size_t) n2 * 8, stream);
                hipLaunchKernelGGL((aligned_tr_scatter_fx_kernel<R>), dim3(P.B), dim3(256), 0, stream, P, W, A, (const R *) gHD, fx);
                hipLaunchKernelGGL((fx_to_grad_kernel<R>), dim3((unsigned) ((n2 + 255) / 256)), dim3(256), 0, stream,
                                   (const unsigned long long *) fx, n2, gtr, have_full ? 1 : 0);
            } else {
                if (!have_full) (void) zero_async(gtr, (size_t) P.N * P.N * e, stream);
                hipLaunchKernelGGL((aligned_tr_scatter_kernel<R>), dim3(1), dim3(1024), 0, stream, P, W, A, gHD, gtr, 1);
            }
        }
    }
    return hipGetLastError();
}

template hipError_t launch_fwd_aligned_generic<float>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);
template hipError_t launch_fwd_aligned_generic<double>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);
template hipError_t launch_bwd_aligned_generic<float>(const Problem &, const State &, const BwdArgs &, const GenericBwdLayout &, bool, bool, hipStream_t);
template hipError_t launch_bwd_aligned_generic<double>(const Problem &, const State &, const BwdArgs &, const GenericBwdLayout &, bool, bool, hipStream_t);

}  // namespace asg
