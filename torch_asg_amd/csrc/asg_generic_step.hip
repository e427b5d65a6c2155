// torch_asg_amd/csrc/asg_generic_step.hip -- generic path, FULL LATTICE FORWARD (N > 64): transition prep, the per-frame step in its
// three forms (LDS-tiled VALU body, 16 x 16 tiles on the matrix instruction of the problem's precision, the fp32 streaming step
// fwd_step_mfma that cfg 5 runs), the medium-alphabet kernel (64 < N <= 256), the resident-slice kernel (256 < N <= 2048) with its
// in-stream repair, the score kernels, and the layout of the forward work area.  The file-level story is in asg_generic.hip.
#include "asg_generic_common.h"

namespace asg {

namespace {

// ------------------------------------------------------------------ transition prep
// grid = N, block = 256.  Row i of out = exp2(Tr2[i][:] - rowmax) (COLS=false) or the same for column i
// (COLS=true: out[i][j] = exp2(Tr2[j][i] - colmax_i)).
template <typename R, bool COLS>
__global__ void __launch_bounds__(256) prep_kernel(const R *tr, int64_t ts0, int64_t ts1, int N, int npad, R *out, R *mx) {
    __shared__ R red[4];
    const int i = blockIdx.x;
    const R L2E = Num<R>::log2e(), NINF = Num<R>::ninf();
    const int64_t sa = COLS ? ts1 : ts0, sb = COLS ? ts0 : ts1;     // element (i,j) at i*sa + j*sb
    R m = NINF;
    for (int j = threadIdx.x; j < N; j += 256) m = fmax(m, tr[(int64_t) i * sa + (int64_t) j * sb] * L2E);
    m = block_reduce_max<R>(m, red);
    if (m == NINF) m = 0;
    for (int j = threadIdx.x; j < npad; j += 256)
        out[(int64_t) i * npad + j] = j < N ? Num<R>::exp2(tr[(int64_t) i * sa + (int64_t) j * sb] * L2E - m) : R(0);
    if (threadIdx.x == 0) mx[i] = m;
}

// The column-normalised twin without strided reads (prep_kernel<R, true> walks a column per workgroup: 16-32x read
// amplification, 18 GB fetched for a 400 MB matrix at N = 10^4).  Two passes over 64 x 64 tiles read along the rows:
//   colmax_kernel:    partial column maxima of a 64-column x 256-row slab -> atomicMax on an order-preserving key
//                     (max is order-independent: deterministic);  keys[] starts at key(-inf)
//   colnorm_kernel:   out[i][j] = exp2(Tr2[j][i] - colmax_i), the tile transposed through LDS so that reads follow
//                     the matrix rows and writes follow the output rows; also writes mx[i] (block row 0)
// grid colmax = (ceil(N/64), ceil(N/256)), colnorm = (ceil(npad/64) over j, ceil(N/64) over i); block = 256.
__device__ __forceinline__ unsigned long long dkey(double f) {
    unsigned long long b = (unsigned long long) __double_as_longlong(f);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long) b);
}
template <typename R>
__global__ void __launch_bounds__(256) colmax_kernel(const R *tr, int64_t ts0, int64_t ts1, int N, unsigned long long *keys) {
    __shared__ R part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rq = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 256;
    R m = Num<R>::ninf();
    if (c < N)
        for (int r = r0 + rq; r < min(r0 + 256, N); r += 4) m = fmax(m, tr[(int64_t) r * ts0 + (int64_t) c * ts1] * Num<R>::log2e());
    part[rq][threadIdx.x & 63] = m;
    __syncthreads();
    if (rq == 0 && c < N) {
        m = fmax(fmax(part[0][threadIdx.x], part[1][threadIdx.x]), fmax(part[2][threadIdx.x], part[3][threadIdx.x]));
        if (m == m) atomicMax(&keys[c], dkey((double) m));
    }
}
template <typename R>
__global__ void __launch_bounds__(256) colnorm_kernel(const R *tr, int64_t ts0, int64_t ts1, int N, int npad,
                                                      const unsigned long long *keys, R *out, R *mx) {
    __shared__ R tile[64][65];
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;          // out rows i0.., out columns j0.. (= matrix rows)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int q = ty; q < 64; q += 4) {                             // matrix row j0 + q, columns i0 + tx: along the rows
        const int j = j0 + q, i = i0 + tx;
        tile[q][tx] = (j < N && i < N) ? tr[(int64_t) j * ts0 + (int64_t) i * ts1] * Num<R>::log2e() : Num<R>::ninf();
    }
    __syncthreads();
    for (int q = ty; q < 64; q += 4) {                             // out row i0 + q, columns j0 + tx
        const int i = i0 + q, j = j0 + tx;
        if (i >= N || j >= npad) continue;
        R m = (R) dunkey(keys[i]);
        if (!(m > Num<R>::ninf())) m = 0;                          // a column of -inf: as prep_kernel
        out[(int64_t) i * npad + j] = j < N ? Num<R>::exp2(tile[tx][q] - m) : R(0);
        if (blockIdx.x == 0 && tx == 0) mx[i] = m;
    }
}

// per-frame emission maximum: emax[t][b] = max_i I2[t][b][i]   (grid = (T, B), block = 256)
template <typename R>
__global__ void __launch_bounds__(256) emax_kernel(Problem P, R *emax) {
    __shared__ R red[4];
    const int t = blockIdx.x, b = blockIdx.y;
    const R *in = (const R *) P.inputs + (int64_t) t * P.is0 + (int64_t) b * P.is1;
    R m = Num<R>::ninf();
    for (int i = threadIdx.x; i < P.N; i += 256) m = fmax(m, in[(int64_t) i * P.is2] * Num<R>::log2e());
    m = block_reduce_max<R>(m, red);
    if (threadIdx.x == 0) emax[(int64_t) t * P.B + b] = fmax(m, Num<R>::logzero());
}

// ------------------------------------------------------------------ forward stepping state
template <typename R>
struct StepBuf {
    R *pbuf;          // [2][B][npad]  p = exp2(q) of the frame being consumed / produced
    unsigned *mu;     // [3][B]        key(max q) per utterance, triple buffered
    double *off;      // [B]           running absolute offset
    const R *emax;    // [T][B]
    R *state;         // ah or bh  [B][T][N]
    const R *ehat;    // [N][npad] (alpha: rows, beta: columns)
    const R *etile;   // fp32: the same matrix in the MFMA step's operand order (tile_kernel)
    const R *hmax;    // [N]
    R *mulog;         // alpha only, [T][B]: the normaliser each frame's state was stored against (read by the gradient pass)
    R *ptile;         // fp32 streaming step only: p again, in the MFMA step's operand order (step_ptile_index), [2][...]; or null
    R *partial;       // fp32 streaming step, K split over several workgroups: [row tile][batch tile][slice][2 MB][256] partial row sums
    unsigned *tickets;   // ... and one arrival counter per (row tile, batch tile), zero at the start of a forward call
    int npad;
    int bf3;             // fp32 streaming step on the bfloat16 pipe (fwd_step_bf3): etile / ptile hold three bfloat16 planes per float
};

// The vectors of the fp32 streaming step in operand order: [batch tile of 32][chunk of 32 k][utterance half u][h][lane][4], lane =
// 16 (k sub-range kq) + (utterance & 15), k = 32 chunk + 8 kq + 4 h + component -- the 16 bytes lane l of a wavefront wants are at
// position l of a contiguous kilobyte, so a wavefront load is eight whole 128-byte lines (row-major vectors: sixteen half lines,
// twice the L2 requests per byte, and it is the L2-resident side traffic that holds the matrix stream back: tools/ubench/stream_side.hip).
__host__ __device__ inline size_t step_ptile_floats(int B, int npad) { return (size_t) ((B + 31) / 32) * ((npad + 31) / 32) * 2 * 2 * 64 * 4; }
__host__ __device__ inline size_t step_ptile_index(int b, int i, int npad) {
    const size_t nch = ((size_t) npad + 31) / 32;
    const int c = i >> 5, kq = (i >> 3) & 3, h = (i >> 2) & 1, u = (b >> 4) & 1;
    return ((((size_t) (b >> 5) * nch + c) * 2 + u) * 2 + h) * 256 + (size_t) (kq * 16 + (b & 15)) * 4 + (i & 3);
}

// ... and as three bfloat16 planes (fwd_step_bf3): [batch tile of 32][chunk of 32 k][utterance half u][plane][lane][8], lane = 16 (k sub-range kq)
// + (utterance & 15), k = 32 chunk + 8 kq + component: the 16 bytes of lane l are the A / B operand of one v_mfma_f32_16x16x32_bf16.
__host__ __device__ inline size_t step_ptile3_elems(int B, int npad) { return (size_t) ((B + 31) / 32) * ((npad + 31) / 32) * 2 * 3 * 64 * 8; }
__host__ __device__ inline size_t step_ptile3_index(int b, int i, int npad, int plane) {
    const size_t nch = ((size_t) npad + 31) / 32;
    const int c = i >> 5, kq = (i >> 3) & 3, u = (b >> 4) & 1;
    return (((((size_t) (b >> 5) * nch + c) * 2 + u) * 3 + plane) * 64 + (size_t) (kq * 16 + (b & 15))) * 8 + (i & 7);
}
// one float as the exact sum of three bfloat16 (split3x2 for a single value)
__device__ __forceinline__ void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
    unsigned a, b, c;
    split3x2(x, 0.f, a, b, c);
    h = (unsigned short) (a & 0xffffu); m = (unsigned short) (b & 0xffffu); l = (unsigned short) (c & 0xffffu);
}
__device__ __forceinline__ void store_ptile3(float *ptile, size_t frame_elems_offset, int b, int i, int npad, float pv) {
    unsigned short h, m, l;
    split3(pv, h, m, l);
    unsigned short *pp = reinterpret_cast<unsigned short *>(ptile) + frame_elems_offset;
    pp[step_ptile3_index(b, i, npad, 0)] = h;
    pp[step_ptile3_index(b, i, npad, 1)] = m;
    pp[step_ptile3_index(b, i, npad, 2)] = l;
}

// init: alpha at frame 0 / beta at frame len-1.  grid = B, block = 256.
template <typename R, bool BETA>
__device__ __forceinline__ void fwd_init_body(const Problem &P, const StepBuf<R> &S, int b) {
    const int N = P.N, T = P.T;
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const R L2E = Num<R>::log2e();
    if (threadIdx.x == 0) {
        S.mu[0 * P.B + b] = fkey(0.0f);
        S.mu[1 * P.B + b] = fkey(-__builtin_inff());
        S.mu[2 * P.B + b] = fkey(-__builtin_inff());
    }
    if (len < 1) { if (threadIdx.x == 0) S.off[b] = -1e300; return; }
    const int t = BETA ? len - 1 : 0;
    const R em = S.emax[(int64_t) t * P.B + b];
    const R *in = (const R *) P.inputs + (int64_t) t * P.is0 + (int64_t) b * P.is1;
    R *st = S.state + ((int64_t) b * T + t) * N;
    R *pb = S.pbuf + (int64_t) b * S.npad;
    for (int i = threadIdx.x; i < S.npad; i += 256) {
        if (i < N) {
            R q = in[(int64_t) i * P.is2] * L2E - em;          // max over i is exactly 0
            st[i] = BETA ? R(0) : q;
            pb[i] = Num<R>::exp2(q);
            if (S.ptile) {      // (frame 0's buffer; pad positions were zeroed by the launcher)
                if (S.bf3) store_ptile3((float *) S.ptile, 0, b, i, S.npad, (float) pb[i]);
                else S.ptile[step_ptile_index(b, i, S.npad)] = pb[i];
            }
        } else {
            pb[i] = 0;
        }
    }
    if (threadIdx.x == 0) S.off[b] = (double) em;
}
template <typename R, bool BETA>
__global__ void __launch_bounds__(256) fwd_init_kernel(Problem P, StepBuf<R> S) { fwd_init_body<R, BETA>(P, S, (int) blockIdx.x); }

// One frame of the recursion for all utterances.  grid = (ceil(N/64), ceil(B/32)), block = 256.
// Thread (r = tid & 63, ug = tid >> 6) owns row i = 64*bx + r and the eight utterances 32*by + 8*ug .. +7, so with
// B <= 32 every tile of E is fetched exactly once per frame.  Tiles are staged global -> registers -> LDS with
// 16-byte loads, the next tile's loads in flight while the current one is consumed.
// step n: alpha consumes q_{t-1} (t = n+1) and writes ah[t]; beta consumes q_t (t = len-1-n) and writes bh[t-1].
template <typename R, bool BETA>
__device__ __forceinline__ void fwd_step_body(const Problem &P, const StepBuf<R> &S, int n) {
    constexpr int KT = 32, NU = 4;
    __shared__ __attribute__((aligned(16))) R Es[KT][64 + 4];
    __shared__ __attribute__((aligned(16))) R Ps[KT][32];
    const int N = P.N, T = P.T, B = P.B, npad = S.npad;
    // thread (rp = tid & 31, ug = tid >> 5) owns rows i0+2rp, i0+2rp+1 and utterances b0+4ug .. +3:
    // per k one ds_read_b64 (two rows) + one ds_read_b128 (four utterances) feed 8 FMAs
    const int rp = threadIdx.x & 31, ug = threadIdx.x >> 5;
    const int i0 = blockIdx.x * 64, b0 = blockIdx.y * 32;
    const R *pcur = S.pbuf + (int64_t) (n & 1) * B * npad;
    R *pnext = S.pbuf + (int64_t) ((n + 1) & 1) * B * npad;

    // staging assignment: E tile = 64 rows x 8 float4 -> 2 float4 per thread; P tile = 32 utt x 8 float4 -> 1
    const int er0 = threadIdx.x >> 3, ec = threadIdx.x & 7;          // rows er0 and er0 + 32, float4 column ec
    const int pu = threadIdx.x >> 3, pc = threadIdx.x & 7;
    const V4<R> zero4 = {0, 0, 0, 0};
    auto ldE = [&](int k0, int rr) -> V4<R> {
        int ii = i0 + rr, jj = k0 + 4 * ec;
        return (ii < N && jj < npad) ? *reinterpret_cast<const V4<R> *>(S.ehat + (int64_t) ii * npad + jj) : zero4;
    };
    auto ldP = [&](int k0) -> V4<R> {
        int bb = b0 + pu, jj = k0 + 4 * pc;
        return (bb < B && jj < npad) ? *reinterpret_cast<const V4<R> *>(pcur + (int64_t) bb * npad + jj) : zero4;
    };
    struct Stage { V4<R> ea, eb, pa; };
    auto fetch = [&](int k0) -> Stage {
        Stage st;
        st.ea = ldE(k0, er0); st.eb = ldE(k0, er0 + 32); st.pa = ldP(k0);
        return st;
    };
    R acc0[NU], acc1[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { acc0[u] = 0; acc1[u] = 0; }
    auto consume = [&](const Stage &st) {
        __syncthreads();
        Es[4 * ec + 0][er0] = st.ea.x; Es[4 * ec + 1][er0] = st.ea.y; Es[4 * ec + 2][er0] = st.ea.z; Es[4 * ec + 3][er0] = st.ea.w;
        Es[4 * ec + 0][er0 + 32] = st.eb.x; Es[4 * ec + 1][er0 + 32] = st.eb.y; Es[4 * ec + 2][er0 + 32] = st.eb.z; Es[4 * ec + 3][er0 + 32] = st.eb.w;
        Ps[4 * pc + 0][pu] = st.pa.x; Ps[4 * pc + 1][pu] = st.pa.y; Ps[4 * pc + 2][pu] = st.pa.z; Ps[4 * pc + 3][pu] = st.pa.w;
        __syncthreads();
    };
    auto compute = [&]() {
#pragma unroll 8
        for (int kk = 0; kk < KT; ++kk) {
            const V2<R> ev = *reinterpret_cast<const V2<R> *>(&Es[kk][2 * rp]);
            const V4<R> q = *reinterpret_cast<const V4<R> *>(&Ps[kk][NU * ug]);
            acc0[0] = fma(ev.x, q.x, acc0[0]); acc0[1] = fma(ev.x, q.y, acc0[1]);
            acc0[2] = fma(ev.x, q.z, acc0[2]); acc0[3] = fma(ev.x, q.w, acc0[3]);
            acc1[0] = fma(ev.y, q.x, acc1[0]); acc1[1] = fma(ev.y, q.y, acc1[1]);
            acc1[2] = fma(ev.y, q.z, acc1[2]); acc1[3] = fma(ev.y, q.w, acc1[3]);
        }
    };
    // two tiles in flight ahead of the one being consumed
    Stage sa = fetch(0), sb = fetch(KT);
    for (int k0 = 0; k0 < npad; k0 += 2 * KT) {
        consume(sa);
        sa = fetch(k0 + 2 * KT);
        compute();
        if (k0 + KT < npad) {
            consume(sb);
            sb = fetch(k0 + 3 * KT);
            compute();
        }
    }
    // ---- epilogue
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int b = b0 + NU * ug + u;            // uniform per half-wave (lanes 0-31 / 32-63 differ in ug)
        const bool bvalid = b < B;
        const int bc = bvalid ? b : 0;
        const int len = P.in_len ? gclampi(P.in_len[bc], 0, T) : T;
        const int t = BETA ? len - 1 - n : n + 1;          // frame whose q is consumed (beta) / produced (alpha)
        const bool active = bvalid && (BETA ? (t >= 1) : (t < len));
        const R muprev = fmax((R) funkey(S.mu[(n % 3) * B + bc]), LZ);
        const int tw = active ? (BETA ? t - 1 : t) : 0;    // frame written
        const R emw = S.emax[(int64_t) tw * B + bc];
        float qkey = -__builtin_inff();
#pragma unroll
        for (int rr2 = 0; rr2 < 2; ++rr2) {
            const int i = i0 + 2 * rp + rr2;
            if (!active || i >= N) continue;
            const R a = rr2 == 0 ? acc0[u] : acc1[u];
            R lg = Num<R>::log2(a);
            R rr = S.hmax[i] + lg;
            if (!(fabs(lg) < Num<R>::lg_limit())) {
                // exact rare path: log2-sum-exp2 over j of (Tr2[.][.] + q_j) from the log-domain state
                const R *tr = (const R *) P.transition;
                const int tq = BETA ? t : t - 1;
                const R *stq = S.state + ((int64_t) b * T + tq) * N;
                const R *inq = (const R *) P.inputs + (int64_t) tq * P.is0 + (int64_t) b * P.is1;
                const R emq = S.emax[(int64_t) tq * B + b];
                R mx = Num<R>::ninf();
                for (int j = 0; j < N; ++j) {
                    R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + stq[j] : stq[j];
                    R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                    R v = trv * L2E + qj;
                    mx = (v == v) ? fmax(mx, v) : mx;
                }
                R sm = 0;
                for (int j = 0; j < N; ++j) {
                    R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + stq[j] : stq[j];
                    R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                    R v = trv * L2E + qj;
                    sm += (v == v && mx != Num<R>::ninf()) ? Num<R>::exp2(v - mx) : R(0);
                }
                rr = (mx == Num<R>::ninf()) ? mx : mx + Num<R>::log2(sm);
            }
            const R emis = ((const R *) P.inputs)[(int64_t) tw * P.is0 + (int64_t) b * P.is1 + (int64_t) i * P.is2] * L2E - emw;
            R stv, q;
            if (BETA) { stv = rr - muprev; q = emis + stv; }
            else { stv = emis + rr - muprev; q = stv; }
            S.state[((int64_t) b * T + tw) * N + i] = stv;
            pnext[(int64_t) b * npad + i] = Num<R>::exp2(q);
            qkey = fmaxf(qkey, (float) q);
            if (i == 0) {
                S.off[b] += (double) muprev + (double) emw;
                S.mu[((n + 2) % 3) * B + b] = fkey(-__builtin_inff());
                if (!BETA && S.mulog) S.mulog[(int64_t) tw * B + b] = muprev;
            }
        }
        // one atomic per half-wave and utterance (max is order-independent: deterministic).  The two halves of a
        // wave hold different utterances, so reduce inside 32-lane halves: four DPP row steps + row_bcast:15.
        qkey = fmaxf(qkey, dpp_mov<kDppXor1>(qkey, qkey));
        qkey = fmaxf(qkey, dpp_mov<kDppXor2>(qkey, qkey));
        qkey = fmaxf(qkey, dpp_mov<kDppHalfMirror>(qkey, qkey));
        qkey = fmaxf(qkey, dpp_mov<kDppMirror>(qkey, qkey));
        const float other = __shfl_xor(qkey, 16);
        qkey = fmaxf(qkey, other);
        if (active && (rp == 0)) atomicMax(&S.mu[((n + 1) % 3) * B + b], fkey(qkey));
    }
}

// ---- the same frame in 16 x 16 tiles on the matrix cores (round 4: fp64 with 256 < N <= 2048; fp32: developer switch only, it
// loses to fwd_step_mfma -- kTileStepMaxN32) ---------------------------------------------------------------------------
// fwd_step_body / fwd_step_mfma are built for N = 10^4 (64- / 80-row output tiles: every tile of E fetched once per frame); at
// N = 512, B = 64 the fp64 body is 32 workgroups of dependent LDS-staged VALU products: 48 us per frame, 19 ms per step where the
// fp32 routes take 1.9.  Here a workgroup owns a 16 x 16 output tile (16 rows, 16 utterances: N / 16 x B / 16 x directions
// workgroups -- 256 at N = 512, B = 64), its four wavefronts a quarter of K each, on v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32.  The order of a dot product's terms is free, so the
// k slot of a lane group is not "k mod 4" but a CONTIGUOUS slice of K: lane (m, kq) of wavefront w walks k = (4 w + kq) KL + j,
// j = 0 .. KL - 1, and loads its row of E and its utterance's vector sixteen bytes at a time, straight from memory (the matrix is
// L2 / memory-side-cache resident at these sizes).  Epilogue: fwd_step_body's, one element per thread.
template <typename R> struct TileOps;
template <> struct TileOps<double> {
    typedef double Ld __attribute__((ext_vector_type(2)));      // one 16-byte load
    typedef V4d Acc;
    static constexpr int EPL = 2;
    static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int q) { return (lane >> 4) + 4 * q; }      // accumulator register q of a lane
};
template <> struct TileOps<float> {
    typedef float Ld __attribute__((ext_vector_type(4)));
    typedef V4<float> Acc;
    static constexpr int EPL = 4;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int q) { return 4 * (lane >> 4) + q; }
};
// NBT = utterance tiles per workgroup (16 NBT utterances share every element of E that is loaded: 2 from N > 512, where the
// matrix no longer sits in the L2 and B / 16 readers per row cost more than the workgroups they add)
template <typename R, bool BETA, int NBT>
__device__ __forceinline__ void fwd_step_tile(const Problem &P, const StepBuf<R> &S, int n, int tile_x, int tile_y) {
    typedef TileOps<R> Ops;
    typedef typename Ops::Ld Ld;
    typedef typename Ops::Acc Acc;
    constexpr int EPL = Ops::EPL;
    __shared__ R red[4][NBT][16][17];
    __shared__ unsigned qk[16 * NBT];
    const int N = P.N, T = P.T, B = P.B, npad = S.npad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int i0 = tile_x * 16, b0 = tile_y * 16 * NBT;
    const R *pcur = S.pbuf + (int64_t) (n & 1) * B * npad;
    R *pnext = S.pbuf + (int64_t) ((n + 1) & 1) * B * npad;
    const int KL = ((npad + 15) / 16 + EPL - 1) / EPL * EPL;   // elements per lane group (whole 16-byte loads)
    const int k0 = (4 * wave + kq) * KL;
    const R *erow = S.ehat + (int64_t) min(i0 + m, N - 1) * npad;
    const R *prow[NBT];
#pragma unroll
    for (int c = 0; c < NBT; ++c) prow[c] = pcur + (int64_t) min(b0 + 16 * c + m, B - 1) * npad;
    if (tid < 16 * NBT) qk[tid] = fkey(-__builtin_inff());
    Acc acc0[NBT], acc1[NBT];
#pragma unroll
    for (int c = 0; c < NBT; ++c) { acc0[c] = Acc{0, 0, 0, 0}; acc1[c] = Acc{0, 0, 0, 0}; }
    constexpr int CH = 8, NL = CH / EPL;                   // elements / 16-byte loads per chunk and operand
    struct Chunk { Ld a[NL], b[NBT][NL]; };
    const Ld zero = {};
    auto fetch = [&](int j0, Chunk &X) {
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            const int k = k0 + j0 + EPL * c;
            const bool in = j0 + EPL * c < KL && k < npad;          // (npad is a multiple of 4: a load is inside or outside as a whole)
            X.a[c] = in ? *reinterpret_cast<const Ld *>(erow + k) : zero;
#pragma unroll
            for (int t = 0; t < NBT; ++t) X.b[t][c] = in ? *reinterpret_cast<const Ld *>(prow[t] + k) : zero;
        }
    };
    Chunk cur, nxt;
    fetch(0, cur);
    for (int j0 = 0; j0 < KL; j0 += CH) {
        fetch(j0 + CH, nxt);
#pragma unroll
        for (int c = 0; c < NL; ++c)
#pragma unroll
            for (int e = 0; e < EPL; e += 2)
#pragma unroll
                for (int t = 0; t < NBT; ++t) {
                    acc0[t] = Ops::mma(cur.a[c][e], cur.b[t][c][e], acc0[t]);
                    acc1[t] = Ops::mma(cur.a[c][e + 1], cur.b[t][c][e + 1], acc1[t]);
                }
        cur = nxt;
    }
    // accumulator register q of lane l = element (row Ops::row(l, q), utterance l & 15)
#pragma unroll
    for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][t][Ops::row(lane, q)][m] = acc0[t][q] + acc1[t][q];
    __syncthreads();
    // ---- epilogue: thread = (row r = tid >> 4, utterance u = tid & 15) of every utterance tile
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero();
    const int r = tid >> 4, i = i0 + r;
#pragma unroll
    for (int ut = 0; ut < NBT; ++ut) {
        const int u = tid & 15, b = b0 + 16 * ut + u;
        const bool bvalid = b < B;
        const int bc = bvalid ? b : 0;
        const int len = P.in_len ? gclampi(P.in_len[bc], 0, T) : T;
        const int t = BETA ? len - 1 - n : n + 1;          // frame whose q is consumed (beta) / produced (alpha)
        const bool active = bvalid && (BETA ? (t >= 1) : (t < len));
        if (active && i < N) {
            const R muprev = fmax((R) funkey(S.mu[(n % 3) * B + bc]), LZ);
            const int tw = BETA ? t - 1 : t;               // frame written
            const R emw = S.emax[(int64_t) tw * B + bc];
            const R a = (red[0][ut][r][u] + red[1][ut][r][u]) + (red[2][ut][r][u] + red[3][ut][r][u]);
            R lg = Num<R>::log2(a);
            R rr = S.hmax[i] + lg;
            if (!(fabs(lg) < Num<R>::lg_limit())) {
                // exact rare path: log2-sum-exp2 over j of (Tr2[.][.] + q_j) from the log-domain state
                const R *tr = (const R *) P.transition;
                const int tq = BETA ? t : t - 1;
                const R *stq = S.state + ((int64_t) b * T + tq) * N;
                const R *inq = (const R *) P.inputs + (int64_t) tq * P.is0 + (int64_t) b * P.is1;
                const R emq = S.emax[(int64_t) tq * B + b];
                R mx = Num<R>::ninf();
                for (int j = 0; j < N; ++j) {
                    R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + stq[j] : stq[j];
                    R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                    R v = trv * L2E + qj;
                    mx = (v == v) ? fmax(mx, v) : mx;
                }
                R sm = 0;
                for (int j = 0; j < N; ++j) {
                    R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + stq[j] : stq[j];
                    R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                    R v = trv * L2E + qj;
                    sm += (v == v && mx != Num<R>::ninf()) ? Num<R>::exp2(v - mx) : R(0);
                }
                rr = (mx == Num<R>::ninf()) ? mx : mx + Num<R>::log2(sm);
            }
            const R emis = ((const R *) P.inputs)[(int64_t) tw * P.is0 + (int64_t) b * P.is1 + (int64_t) i * P.is2] * L2E - emw;
            R stv, q;
            if (BETA) { stv = rr - muprev; q = emis + stv; }
            else { stv = emis + rr - muprev; q = stv; }
            S.state[((int64_t) b * T + tw) * N + i] = stv;
            pnext[(int64_t) b * npad + i] = Num<R>::exp2(q);
            // the utterance's maximum of q over this tile's rows (max is order-independent: deterministic)
            atomicMax(&qk[16 * ut + u], fkey((float) q));
            if (i == 0) {
                S.off[b] += (double) muprev + (double) emw;
                S.mu[((n + 2) % 3) * B + b] = fkey(-__builtin_inff());
                if (!BETA && S.mulog) S.mulog[(int64_t) tw * B + b] = muprev;
            }
        }
    }
    __syncthreads();
    if (tid < 16 * NBT) {        // one global atomic per tile and utterance
        const int bb = b0 + tid;
        const unsigned key = qk[tid];
        if (bb < B && key != fkey(-__builtin_inff())) atomicMax(&S.mu[((n + 1) % 3) * B + bb], key);
    }
}
template <typename R, int NBT>
__global__ void __launch_bounds__(256) fwd_step_tile_kernel(Problem P, StepBuf<R> Sa, StepBuf<R> Sb, int n, int dir_base) {
    if ((int) blockIdx.z + dir_base == 0) fwd_step_tile<R, false, NBT>(P, Sa, n, (int) blockIdx.x, (int) blockIdx.y);
    else fwd_step_tile<R, true, NBT>(P, Sb, n, (int) blockIdx.x, (int) blockIdx.y);
}

// ---- in-stream repair of a resident-slice launch that timed out (round 6) -------------------------------------------------
// fwd_cluster_kernel's workgroups wait for each other; a wait that runs out used to poison the scores with NaN and the error surfaced at
// the NEXT call -- after an optimizer may have stepped.  Now the launch that timed out raises a per-call word in its own work area, and
// this kernel, enqueued behind it on the same stream by the same call, reads that word: zero (every launch but a faulted one) and it
// returns at once (~2 us per call of a >= 1.8 ms route); non-zero and it REDOES the whole full-lattice recursion of the call with no
// dependence between workgroups -- one workgroup per 16 utterances and direction, initial state as fwd_init_kernel, then frame after frame
// the 16 x 16 tile step of the launch-per-frame route (fwd_step_tile: the same arithmetic, states and scale log the gradient pass reads)
// over all row tiles, agent-scope release / acquire around a barrier between frames.  Slow (tens of milliseconds) and exact; the
// process-wide count (asg_cluster_timeouts) still makes later calls take the launch-per-frame kernels.
template <typename R>
__global__ void __launch_bounds__(256) fwd_repair_kernel(Problem P, StepBuf<R> Sa, StepBuf<R> Sb, const unsigned *callfault, int dir_base) {
    if (__hip_atomic_load(callfault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const bool beta = (int) blockIdx.y + dir_base != 0;
    const StepBuf<R> &S = beta ? Sb : Sa;
    const int by = (int) blockIdx.x, tiles = (P.N + 15) / 16;
    for (int u = 0; u < 16; ++u) {
        const int b = by * 16 + u;
        if (b >= P.B) break;
        if (beta) fwd_init_body<R, true>(P, S, b);
        else fwd_init_body<R, false>(P, S, b);
    }
    for (int n = 0; n + 1 < P.T; ++n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int x = 0; x < tiles; ++x) {
            if (beta) fwd_step_tile<R, true, 1>(P, S, n, x, by);
            else fwd_step_tile<R, false, 1>(P, S, n, x, by);
        }
    }
}

// ---- the same frame on the matrix cores (fp32 only) ------------------------------------------------------------
// The large-alphabet step IS a dense product, [N x N] (normalised transitions) x [N x B] (the batch's vectors), 8 FMAs
// per byte of E streamed: at B = 32 the HBM and the fp32 arithmetic ceilings of MI355X coincide (~100 us per frame and
// direction pair at N = 10^4).  The LDS-staged VALU body above reaches a fifth of either (LDS operand traffic: one
// ds_read_b64 + one ds_read_b128 per 8 FMAs).  v_mfma_f32_16x16x4_f32 is exact fp32 (a k-ordered fmaf chain) at the
// vector peak rate and takes its operands straight from the registers the global loads fill:
//   workgroup = 16 MB rows of E (MB = 2 .. kStepMB row blocks) x 32 utterances, K split over its 4 wavefronts (partial sums meet in LDS)
//   lane l of a wavefront loads E[row i0 + (l & 15)][k + 8 (l >> 4) .. +7] (two float4; a row's 128 contiguous bytes per
//   32 k) and V[utterance (l & 15) (+16)][same k]; component c of one of those float4 is the A / B operand of one MFMA:
//   A[m = l & 15][kk = l >> 4], B[kk = l >> 4][n = l & 15] -- the four k of an instruction are {c, 8+c, 16+c, 24+c}
//   (+4 for the second float4), the same set on both sides, which is all the contraction needs.
// grid = row tiles x slices of K x directions x batch tile groups, one-dimensional and XCD-aware (fwd_step_kernel; the shape is
// step_plan's and step_slices' choice).  Tile height decides two things: every workgroup reads the batch's whole vector set
// (1.3 MB at cfg 5, from L2), and the workgroup count has to divide evenly over 256 compute units.
// Measured at cfg 5 (us per frame, both directions; tools/cfg5_fwd_time.py): 16 rows 356 (1250 workgroups, 3.2 GB of
// vectors per frame) - 48 rows 213 (418 workgroups = 1.6 per compute unit: half the chip waits for the other half) -
// 80 rows 151 (250 workgroups, one per compute unit) - 96 rows 161.  The VALU body above: 509 (314 workgroups).
#ifndef ASG_X_STEP_PF
#define ASG_X_STEP_PF 1
#endif
#ifndef ASG_X_STEP_PF_MB2
#define ASG_X_STEP_PF_MB2 1
#endif
#ifndef ASG_X_STEP_PF_MB3
#define ASG_X_STEP_PF_MB3 1
#endif
#ifndef ASG_X_STEP_PF_MB4
#define ASG_X_STEP_PF_MB4 1
#endif
constexpr int kStepMB = 5;                  // 16-row blocks per workgroup, at most: every workgroup reads the batch's whole vector
                                            // set once (L2 traffic = row tiles x 1.3 MB), so tiles must not be too small.  2 .. 4
                                            // where that fills the device better (step_plan: N = 3000 at B = 64 is 152 workgroups
                                            // of 80 rows on 256 compute units, 252 of 48 rows)
// The MFMA step's E operand, laid out so that every wavefront load is ONE contiguous kilobyte and a workgroup streams
// its K quarter front to back: [row tile of 16 kStepMB rows][chunk of 32 k][row block m][half h][lane][4 floats], lane l =
// row (l & 15) of the block, k = 32 chunk + 8 (l >> 4) + 4 h .. +3.  Row-major E handed the memory system 16 kStepMB x 4
// interleaved 128-byte streams per workgroup (80 000 on the chip): 3 TB/s; zero-padded to whole tiles and chunks.
__host__ __device__ inline size_t step_tile_floats(int N, int mb) {
    const size_t npad = (size_t) (N + 3) / 4 * 4;
    const size_t tiles = ((size_t) N + 16 * mb - 1) / (16 * mb), chunks = (npad + 31) / 32;
    return tiles * chunks * mb * 2 * 64 * 4;
}
// (what the buffer is sized for: any tile height up to kStepMB)
__host__ __device__ inline size_t step_tile_floats_max(int N) {
    const size_t npad = (size_t) (N + 3) / 4 * 4;
    return (((size_t) N + 15) / 16 + kStepMB) * ((npad + 31) / 32) * 2 * 64 * 4;
}
__global__ void __launch_bounds__(256) tile_kernel(const float *src, int N, int npad, int mb, float *dst) {
    const size_t chunks = ((size_t) npad + 31) / 32;
    const size_t total = step_tile_floats(N, mb) / 4;             // float4 elements
    for (size_t idx = (size_t) blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t) gridDim.x * 256) {
        const int lane = (int) (idx & 63);
        size_t rest = idx >> 6;
        const int h = (int) (rest & 1); rest >>= 1;
        const int m = (int) (rest % mb); rest /= mb;
        const size_t c = rest % chunks, tile = rest / chunks;
        const size_t row = tile * 16 * mb + 16 * m + (lane & 15), k = 32 * c + 8 * (lane >> 4) + 4 * h;
        V4f v = {0, 0, 0, 0};
        if (row < (size_t) N && k < (size_t) npad) v = *reinterpret_cast<const V4f *>(src + row * npad + k);
        reinterpret_cast<V4f *>(dst)[idx] = v;
    }
}


// The same operand for the bfloat16 pipe (fwd_step_bf3): every float as three bfloat16 planes (exact: 8 + 8 + 8 significant bits), [row
// tile][chunk of 32 k][row block m][plane][lane][8 bfloat16], lane l = row (l & 15) of the block, k = 32 chunk + 8 (l >> 4) .. +7 -- the 16
// bytes of a lane are the A operand of one v_mfma_f32_16x16x32_bf16; 1.5x the bytes of the fp32 tile.
__host__ __device__ inline size_t step_tile3_units(int N, int mb) {          // 16-byte units
    const size_t npad = (size_t) (N + 3) / 4 * 4;
    const size_t tiles = ((size_t) N + 16 * mb - 1) / (16 * mb), chunks = (npad + 31) / 32;
    return tiles * chunks * mb * 3 * 64;
}
__host__ __device__ inline size_t step_tile3_units_max(int N) {
    const size_t npad = (size_t) (N + 3) / 4 * 4;
    return (((size_t) N + 15) / 16 + kStepMB) * ((npad + 31) / 32) * 3 * 64;
}
__global__ void __launch_bounds__(256) tile3_kernel(const float *src, int N, int npad, int mb, U4v *dst) {
    const size_t chunks = ((size_t) npad + 31) / 32;
    const size_t total = step_tile3_units(N, mb) / 3;             // (tile, chunk, m, lane) quadruples: three units each
    for (size_t idx = (size_t) blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t) gridDim.x * 256) {
        const int lane = (int) (idx & 63);
        size_t rest = idx >> 6;
        const int m = (int) (rest % mb); rest /= mb;
        const size_t c = rest % chunks, tile = rest / chunks;
        const size_t row = tile * 16 * mb + 16 * m + (lane & 15), k = 32 * c + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (row < (size_t) N && k + q < (size_t) npad) ? src[row * npad + k + q] : 0.f;
        U4v h, md, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned a, b_, c_;
            split3x2(v[2 * q], v[2 * q + 1], a, b_, c_);
            h[q] = a; md[q] = b_; l[q] = c_;
        }
        U4v *o = dst + (((tile * chunks + c) * mb + m) * 3) * 64 + lane;
        o[0] = h; o[64] = md; o[128] = l;
    }
}

// What the frame's epilogue needs from memory, requested BEFORE the product: thread -> utterance 32 bt + (tid >> 3), rows
// i0 + 16 (rr2 >> 1) + 2 (tid & 7) + (rr2 & 1).  (Loaded inside the epilogue, the 2 MB emission values of a thread -- each a miss all the
// way to memory, behind a rarely taken branch hipcc will not load across -- cost the epilogue 7 of its 9 us at cfg 5.)
template <int MB>
struct StepPre {
    int b, len, t, tw;
    bool active;
    float muprev, emw;
    float x[2 * MB], hm[2 * MB];          // raw emission at frame tw, hmax, of the thread's rows (0 where there is no row)
};
template <bool BETA, int MB>
__device__ __forceinline__ StepPre<MB> step_prefetch(const Problem &P, const StepBuf<float> &S, int n, int row_tile, int bt) {
    const int N = P.N, T = P.T, B = P.B;
    const int i0 = row_tile * (16 * MB);
    StepPre<MB> E;
    E.b = bt * 32 + (int) (threadIdx.x >> 3);
    const bool bvalid = E.b < B;
    const int bc = bvalid ? E.b : 0;
    E.len = P.in_len ? gclampi(P.in_len[bc], 0, T) : T;
    E.t = BETA ? E.len - 1 - n : n + 1;          // frame whose q is consumed (beta) / produced (alpha)
    E.active = bvalid && (BETA ? (E.t >= 1) : (E.t < E.len));
    E.muprev = fmax(funkey(S.mu[(n % 3) * B + bc]), Num<float>::logzero());
    E.tw = E.active ? (BETA ? E.t - 1 : E.t) : 0;    // frame written
    E.emw = S.emax[(int64_t) E.tw * B + bc];
    const float *in = (const float *) P.inputs + (int64_t) E.tw * P.is0 + (int64_t) bc * P.is1;
#pragma unroll
    for (int rr2 = 0; rr2 < 2 * MB; ++rr2) {
        const int i = i0 + 16 * (rr2 >> 1) + 2 * (int) (threadIdx.x & 7) + (rr2 & 1);
        const bool on = E.active && i < N;
        const int ic = on ? i : 0;
        const float xv = in[(int64_t) ic * P.is2], hv = S.hmax[ic];
        E.x[rr2] = on ? xv : 0.f;
        E.hm[rr2] = on ? hv : 0.f;
    }
    return E;
}

// The frame's epilogue: red[w] holds wavefront w's partial tile -- element (row 16 m + 4 (l >> 4) + q, utterance (l & 15) [+ 16]) in
// red[w][8 m + q (+ 4)][l].  With K split over ks workgroups (slice = this workgroup's), every workgroup leaves its 2 MB sums per
// thread in S.partial (write-through) and takes a ticket; the LAST one to arrive adds the slices in ascending order (its own from
// registers, in its place: a fixed order whoever is last, so the result does not depend on the arrival order) and runs the epilogue.
// Nobody waits for anybody.  Returns without doing anything in the workgroups that were not last.
// NB batch tiles per workgroup: red[w][j MB 8 + ...] is batch tile j's part; nbt = batch tiles of the problem.
template <bool BETA, int NB, int MB>
__device__ __forceinline__ void step_epilogue(const Problem &P, const StepBuf<float> &S, int n, const float (*red)[NB * MB * 8][64], int j,
                                              const StepPre<MB> &E, int row_tile, int bt, int nbt, int slice, int ks) {
    typedef float R;
    const int N = P.N, T = P.T, B = P.B, npad = S.npad;
    const int i0 = row_tile * (16 * MB);
    R *pnext = S.pbuf + (int64_t) ((n + 1) & 1) * B * npad;
    const int ut = threadIdx.x >> 3, b = E.b;
    R sums[2 * MB];
#pragma unroll
    for (int rr2 = 0; rr2 < 2 * MB; ++rr2) {
        const int row = 16 * (rr2 >> 1) + 2 * (threadIdx.x & 7) + (rr2 & 1);
        const int sl = 16 * ((row & 15) >> 2) + (ut & 15), sq = j * (MB * 8) + 8 * (row >> 4) + (row & 3) + 4 * (ut >> 4);
        sums[rr2] = (red[0][sq][sl] + red[1][sq][sl]) + (red[2][sq][sl] + red[3][sq][sl]);
    }
    if (ks > 1) {
        __shared__ int last_arrival;
        const size_t tile = (size_t) row_tile * nbt + bt;
        R *mine = S.partial + ((tile * ks + slice) * (2 * MB)) * 256 + threadIdx.x;
#pragma unroll
        for (int rr2 = 0; rr2 < 2 * MB; ++rr2) __hip_atomic_store(mine + rr2 * 256, sums[rr2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (no release fence: an agent-scope release writes back every dirty line of this XCD's L2 -- the frame's state and vector
        // stores of 32 workgroups.  The partial sums are write-through stores, drained here.)
#if defined(ASG_X_KS_SYNC) && ASG_X_KS_SYNC == 1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(&S.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_arrival = (t % (unsigned) ks) == (unsigned) (ks - 1);
        }
        __syncthreads();
        if (!last_arrival) return;
        // (acquire: invalidates what this XCD's L2 holds of other XCDs' memory -- a slice written in an earlier frame may still be there)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        R total[2 * MB];
#pragma unroll
        for (int rr2 = 0; rr2 < 2 * MB; ++rr2) total[rr2] = 0;
        for (int k = 0; k < ks; ++k) {
            const R *theirs = S.partial + ((tile * ks + k) * (2 * MB)) * 256 + threadIdx.x;
#pragma unroll
            for (int rr2 = 0; rr2 < 2 * MB; ++rr2) {
                const R v = (k == slice) ? sums[rr2] : __hip_atomic_load(theirs + rr2 * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                total[rr2] = k == 0 ? v : total[rr2] + v;
            }
        }
#pragma unroll
        for (int rr2 = 0; rr2 < 2 * MB; ++rr2) sums[rr2] = total[rr2];
    }
    const R L2E = Num<R>::log2e();
    const bool active = E.active;
    const int t = E.t, tw = E.tw;
    const R muprev = E.muprev, emw = E.emw;
    float qkey = -__builtin_inff();
#pragma unroll
    for (int rr2 = 0; rr2 < 2 * MB; ++rr2) {
        const int row = 16 * (rr2 >> 1) + 2 * (threadIdx.x & 7) + (rr2 & 1), i = i0 + row;
        if (!active || i >= N) continue;
        const R a = sums[rr2];
        R lg = Num<R>::log2(a);
        R rr = E.hm[rr2] + lg;
        if (!(fabs(lg) < Num<R>::lg_limit())) {
            // exact rare path: log2-sum-exp2 over j of (Tr2[.][.] + q_j) from the log-domain state
            const R *tr = (const R *) P.transition;
            const int tq = BETA ? t : t - 1;
            const R *stq = S.state + ((int64_t) b * T + tq) * N;
            const R *inq = (const R *) P.inputs + (int64_t) tq * P.is0 + (int64_t) b * P.is1;
            const R emq = S.emax[(int64_t) tq * B + b];
            R mx = Num<R>::ninf();
            for (int j = 0; j < N; ++j) {
                R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + stq[j] : stq[j];
                R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                R v = trv * L2E + qj;
                mx = (v == v) ? fmax(mx, v) : mx;
            }
            R sm = 0;
            for (int j = 0; j < N; ++j) {
                R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + stq[j] : stq[j];
                R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                R v = trv * L2E + qj;
                sm += (v == v && mx != Num<R>::ninf()) ? Num<R>::exp2(v - mx) : R(0);
            }
            rr = (mx == Num<R>::ninf()) ? mx : mx + Num<R>::log2(sm);
        }
        const R emis = E.x[rr2] * L2E - emw;
        R stv, q;
        if (BETA) { stv = rr - muprev; q = emis + stv; }
        else { stv = emis + rr - muprev; q = stv; }
        const R pv = Num<R>::exp2(q);
        S.state[((int64_t) b * T + tw) * N + i] = stv;
        pnext[(int64_t) b * npad + i] = pv;
        if (S.bf3) store_ptile3(S.ptile, (size_t) ((n + 1) & 1) * step_ptile3_elems(B, npad), b, i, npad, pv);
        else S.ptile[(size_t) ((n + 1) & 1) * step_ptile_floats(B, npad) + step_ptile_index(b, i, npad)] = pv;
        qkey = fmaxf(qkey, (float) q);
        if (i == 0) {
            S.off[b] += (double) muprev + (double) emw;
            S.mu[((n + 2) % 3) * B + b] = fkey(-__builtin_inff());
            if (!BETA && S.mulog) S.mulog[(int64_t) tw * B + b] = muprev;
        }
    }
    // one atomic per utterance and workgroup at most (max is order-independent: deterministic), and only if it can
    // change the word: the eight lanes of an utterance reduce with three DPP steps
    qkey = fmaxf(qkey, dpp_mov<kDppXor1>(qkey, qkey));
    qkey = fmaxf(qkey, dpp_mov<kDppXor2>(qkey, qkey));
    qkey = fmaxf(qkey, dpp_mov<kDppHalfMirror>(qkey, qkey));
    if (active && (threadIdx.x & 7) == 0) {
        unsigned *word = &S.mu[((n + 1) % 3) * B + b];
        const unsigned key = fkey(qkey);
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < key) atomicMax(word, key);
    }
}

#ifdef ASG_X_STEP_PROBE
__device__ long long g_step_probe[4096];
#endif
// NB batch tiles of 32 utterances per workgroup: every element of the matrix tile read from memory multiplies NB * 32 utterances
// (the matrix is streamed ONCE per frame and direction for both; whether that beats one batch tile per workgroup -- whose siblings share
// the tile through the L2 -- is step_plan's pricing).  nbt = batch tiles of the problem.
// NT: the matrix loads are non-temporal (the matrices do not fit the 256 MB memory-side cache and no other workgroup wants the same
// tile) or take the default policy (they fit and stay there from frame to frame, or sibling workgroups share the tile through the L2).
// HALF: the batch has at most 16 utterances -- the second half of the 32-utterance tile does not exist, its vector loads and matrix
// instructions are not issued (the epilogue's threads of those utterances are inactive anyway).
template <bool BETA, int NB, int MB, bool NT, bool HALF>
__device__ __forceinline__ void fwd_step_mfma(const Problem &P, const StepBuf<float> &S, int n, float (*red)[NB * MB * 8][64], int ks, int nbt,
                                              int row_tile, int group, int slice) {
    const int B = P.B, npad = S.npad;
    // (readfirstlane: the chunk index derives from the wavefront's number and has to be a scalar for the buffer loads' offsets --
    // "threadIdx.x >> 6" alone is not provably uniform, and a vector offset turns every load into a waterfall loop)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int bt0 = group * NB;          // first batch tile of 32 utterances
    StepPre<MB> pre[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) pre[j] = step_prefetch<BETA, MB>(P, S, n, row_tile, bt0 + j);
    {
        // lane l: row / utterance (l & 15), k sub-range 8 (l >> 4) .. +7 of every 32-k chunk: two float4 per operand, so a
        // row's whole 128-byte line goes to one wavefront at once
        const size_t nchunks = ((size_t) npad + 31) / 32;
        const V4f *et = reinterpret_cast<const V4f *>(S.etile) + (size_t) row_tile * nchunks * (MB * 2 * 64) + lane;
        // the vectors in operand order (step_ptile_index): kilobyte (chunk, utterance half, h) of a batch tile, position `lane`
        const V4f *pt = reinterpret_cast<const V4f *>(S.ptile + (size_t) (n & 1) * step_ptile_floats(B, npad)) + lane;
        // K in chunks of 32 (the matrix tile and the vectors are zero-padded to whole chunks: every load is unconditional -- a
        // bounds test per load makes hipcc wait for every load before the first MFMA): this workgroup's slice, a quarter of it per
        // wavefront, through a two-stage software pipeline
        const int per = ((int) nchunks + ks - 1) / ks, s0 = min(slice * per, (int) nchunks), s1 = min(s0 + per, (int) nchunks);
        const int cpw = (s1 - s0 + 3) / 4;
        const int c0 = min(s0 + wave * cpw, s1), c1 = min(c0 + cpw, s1);
        const V4f zero4 = {0, 0, 0, 0};
        V4f acc[NB][MB][2];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) { acc[j][m][0] = zero4; acc[j][m][1] = zero4; }
        struct Stage { V4f e[MB][2], a[NB][2], b[NB][2]; };
#ifndef ASG_X_STEP_BUFLOAD
#define ASG_X_STEP_BUFLOAD 1          // 1: raw buffer loads (descriptor + chunk offset in scalar registers, the lane's 16 l bytes in ONE vector register)
#endif
        // this workgroup's matrix tile and its batch tiles' vectors as buffer resources: a load then names a scalar chunk offset and the
        // SAME vector register every time -- no 64-bit address per lane and load.  (A batch tile past the last one: a resource of no
        // records, whose loads return zeros.)
        __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void *) (et - lane), 0, (unsigned) (nchunks * (MB * 2 * 64) * 16), 0x00020000);
        __amdgpu_buffer_rsrc_t rsP[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool there = bt0 + j < nbt;
            rsP[j] = __builtin_amdgcn_make_buffer_rsrc((void *) (pt - lane + (size_t) (there ? bt0 + j : bt0) * nchunks * 256), 0,
                                                       there ? (unsigned) (nchunks * 256 * 16) : 0u, 0x00020000);
        }
        const unsigned vlane = (unsigned) lane * 16u;
        typedef unsigned RawU4 __attribute__((ext_vector_type(4)));
        auto load = [&](Stage &st, int c) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (ASG_X_STEP_BUFLOAD) {
                    const unsigned cp = (unsigned) c * 4096u, ce = (unsigned) c * (MB * 2 * 1024u);
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        st.a[j][h] = __builtin_bit_cast(V4f, (RawU4) __builtin_amdgcn_raw_buffer_load_b128(rsP[j], vlane, cp + h * 1024u, 0));
                        if (!HALF) st.b[j][h] = __builtin_bit_cast(V4f, (RawU4) __builtin_amdgcn_raw_buffer_load_b128(rsP[j], vlane, cp + (2 + h) * 1024u, 0));
                    }
#pragma unroll
                    for (int m = 0; m < MB; ++m)          // (aux 2 = non-temporal: see below)
                        st.e[m][h] = __builtin_bit_cast(V4f, (RawU4) __builtin_amdgcn_raw_buffer_load_b128(rsE, vlane, ce + (m * 2 + h) * 1024u, NT ? 2 : 0));
                    continue;
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const V4f *ptj = pt + (size_t) min(bt0 + j, nbt - 1) * nchunks * 256;
                    st.a[j][h] = ptj[((size_t) c * 4 + h) * 64];
                    if (!HALF) st.b[j][h] = ptj[((size_t) c * 4 + 2 + h) * 64];
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    // (non-temporal: every element of the matrix is used once per frame, and the lines it would displace in
                    // L2 are the batch's vectors that all workgroups of the XCD read: 144.4 -> 137.9 us per frame at cfg 5)
                    st.e[m][h] = NT ? __builtin_nontemporal_load(&et[((size_t) c * MB * 2 + m * 2 + h) * 64]) : et[((size_t) c * MB * 2 + m * 2 + h) * 64];
                }
            }
        };
        auto multiply = [&](const Stage &st) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc[j][m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].x, st.a[j][h].x, acc[j][m][0], 0, 0, 0);
                        if (!HALF) acc[j][m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].x, st.b[j][h].x, acc[j][m][1], 0, 0, 0);
                        acc[j][m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].y, st.a[j][h].y, acc[j][m][0], 0, 0, 0);
                        if (!HALF) acc[j][m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].y, st.b[j][h].y, acc[j][m][1], 0, 0, 0);
                        acc[j][m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].z, st.a[j][h].z, acc[j][m][0], 0, 0, 0);
                        if (!HALF) acc[j][m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].z, st.b[j][h].z, acc[j][m][1], 0, 0, 0);
                        acc[j][m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].w, st.a[j][h].w, acc[j][m][0], 0, 0, 0);
                        if (!HALF) acc[j][m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.e[m][h].w, st.b[j][h].w, acc[j][m][1], 0, 0, 0);
                    }
        };
        if (c0 < c1) {
            // STG stages: STG - 1 chunks of loads in flight while one is multiplied.  (A compute unit holds one of these
            // workgroups = one wavefront per SIMD: the depth has to come from the pipeline.)
            constexpr int STG = (MB == 2 ? ASG_X_STEP_PF_MB2 : MB == 3 ? ASG_X_STEP_PF_MB3 : MB == 4 ? ASG_X_STEP_PF_MB4 : ASG_X_STEP_PF) + 1;
            Stage st[STG];
#pragma unroll
            for (int u = 0; u < STG - 1; ++u) {
                __builtin_amdgcn_sched_barrier(0);
                load(st[u], min(c0 + u, c1 - 1));
            }
            for (int c = c0; c < c1; c += STG) {
#pragma unroll
                for (int u = 0; u < STG; ++u) {
                    // (pinned: left alone, the scheduler sinks each stage's loads next to their MFMAs and the pipeline is gone)
                    __builtin_amdgcn_sched_barrier(0);
                    load(st[(u + STG - 1) % STG], min(c + u + STG - 1, c1 - 1));
                    __builtin_amdgcn_sched_barrier(0);
                    if (c + u < c1) multiply(st[u]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // element (row 16 m + 4 (l >> 4) + q, utterance (l & 15) [+ 16]) of batch tile j's tile sits in register q of lane l
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    red[wave][j * (MB * 8) + 8 * m + q][lane] = acc[j][m][0][q];
                    red[wave][j * (MB * 8) + 8 * m + 4 + q][lane] = acc[j][m][1][q];
                }
    }
    __syncthreads();
#ifdef ASG_X_STEP_PROBE
    if (n == 20 && threadIdx.x == 0) { const int wg = blockIdx.x; if (wg < 1024) g_step_probe[wg * 4 + 1] = wall_clock64(); }
#endif
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (bt0 + j < nbt) step_epilogue<BETA, NB, MB>(P, S, n, red, j, pre[j], row_tile, bt0 + j, nbt, slice, ks);
}



// ---- the same frame on the BFLOAT16 matrix pipe at fp32 accuracy (round 6) ---------------------------------------------------------
// v_mfma_f32_16x16x4_f32 is exact but runs at the vector rate: from 48 utterances up the fp32 step is bound by the issue of its matrix
// instructions (49 cycles apiece fed from memory), not by the matrix stream.  Every float is the exact sum of three bfloat16 (8 + 8 + 8
// significant bits); six partial products hh, hm, mh, hl, lh, mm on v_mfma_f32_16x16x32_bf16 (fp32 accumulation; what is dropped is below
// 2^-24 of the product: the arithmetic of the gradient contraction bwd_gemm_bf3_kernel) do the work of eight fp32 instructions in ~a
// third of the cycles.  The matrix is split ONCE per call (tile3_kernel: 1.5x the bytes), the frame's vectors are written as planes by the
// epilogue that produces them -- no conversion instruction sits in the product loop (round 5's bf16x3 step converted the fp32 matrix in
// registers every frame and lost to its own conversions: tools/experiments/README.md).  Same work split (a quarter of K per wavefront, partial
// tiles meet in `red`), same accumulator layout, same epilogue as fwd_step_mfma.
template <bool BETA, int NB, int MB, bool NT>
__device__ __forceinline__ void fwd_step_bf3(const Problem &P, const StepBuf<float> &S, int n, float (*red)[NB * MB * 8][64], int ks, int nbt,
                                             int row_tile, int group, int slice) {
    const int B = P.B, npad = S.npad;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int bt0 = group * NB;
    StepPre<MB> pre[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) pre[j] = step_prefetch<BETA, MB>(P, S, n, row_tile, bt0 + j);
    {
        const size_t nchunks = ((size_t) npad + 31) / 32;
        const U4v *et = reinterpret_cast<const U4v *>(S.etile) + (size_t) row_tile * nchunks * (MB * 3 * 64);
        const U4v *pt = reinterpret_cast<const U4v *>(reinterpret_cast<const unsigned short *>(S.ptile) + (size_t) (n & 1) * step_ptile3_elems(B, npad));
        const int per = ((int) nchunks + ks - 1) / ks, s0 = min(slice * per, (int) nchunks), s1 = min(s0 + per, (int) nchunks);
        const int cpw = (s1 - s0 + 3) / 4;
        const int c0 = min(s0 + wave * cpw, s1), c1 = min(c0 + cpw, s1);
        const V4f zero4 = {0, 0, 0, 0};
        V4f acc[NB][MB][2];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m) { acc[j][m][0] = zero4; acc[j][m][1] = zero4; }
        struct Stage { BF8 e[MB][3], a[NB][3], b[NB][3]; };
        __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void *) et, 0, (unsigned) (nchunks * (MB * 3 * 64) * 16), 0x00020000);
        __amdgpu_buffer_rsrc_t rsP[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool there = bt0 + j < nbt;
            rsP[j] = __builtin_amdgcn_make_buffer_rsrc((void *) (pt + (size_t) (there ? bt0 + j : bt0) * nchunks * (2 * 3 * 64)), 0,
                                                       there ? (unsigned) (nchunks * (2 * 3 * 64) * 16) : 0u, 0x00020000);
        }
        const unsigned vlane = (unsigned) lane * 16u;
        typedef unsigned RawU4 __attribute__((ext_vector_type(4)));
        auto load = [&](Stage &st, int c) {
            const unsigned cp = (unsigned) c * (2 * 3 * 1024u), ce = (unsigned) c * (MB * 3 * 1024u);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    st.a[j][pl] = __builtin_bit_cast(BF8, (RawU4) __builtin_amdgcn_raw_buffer_load_b128(rsP[j], vlane, cp + pl * 1024u, 0));
                    st.b[j][pl] = __builtin_bit_cast(BF8, (RawU4) __builtin_amdgcn_raw_buffer_load_b128(rsP[j], vlane, cp + (3 + pl) * 1024u, 0));
                }
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    st.e[m][pl] = __builtin_bit_cast(BF8, (RawU4) __builtin_amdgcn_raw_buffer_load_b128(rsE, vlane, ce + (m * 3 + pl) * 1024u, NT ? 2 : 0));
            }
        };
        auto multiply = [&](const Stage &st) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    // (smallest terms first)
#define ASG_BF3_SIX(ACC, V) \
                    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.e[m][2], V[0], ACC, 0, 0, 0); \
                    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.e[m][0], V[2], ACC, 0, 0, 0); \
                    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.e[m][1], V[1], ACC, 0, 0, 0); \
                    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.e[m][1], V[0], ACC, 0, 0, 0); \
                    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.e[m][0], V[1], ACC, 0, 0, 0); \
                    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.e[m][0], V[0], ACC, 0, 0, 0);
                    ASG_BF3_SIX(acc[j][m][0], st.a[j])
                    ASG_BF3_SIX(acc[j][m][1], st.b[j])
#undef ASG_BF3_SIX
                }
        };
        if (c0 < c1) {
            constexpr int STG = 3;          // two chunks of loads in flight while one is multiplied
            Stage st[STG];
#pragma unroll
            for (int u = 0; u < STG - 1; ++u) {
                __builtin_amdgcn_sched_barrier(0);
                load(st[u], min(c0 + u, c1 - 1));
            }
            for (int c = c0; c < c1; c += STG) {
#pragma unroll
                for (int u = 0; u < STG; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
                    load(st[(u + STG - 1) % STG], min(c + u + STG - 1, c1 - 1));
                    __builtin_amdgcn_sched_barrier(0);
                    if (c + u < c1) multiply(st[u]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // element (row 16 m + 4 (l >> 4) + q, utterance (l & 15) [+ 16]) of batch tile j's tile sits in register q of lane l: as fwd_step_mfma
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    red[wave][j * (MB * 8) + 8 * m + q][lane] = acc[j][m][0][q];
                    red[wave][j * (MB * 8) + 8 * m + 4 + q][lane] = acc[j][m][1][q];
                }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (bt0 + j < nbt) step_epilogue<BETA, NB, MB>(P, S, n, red, j, pre[j], row_tile, bt0 + j, nbt, slice, ks);
}
// grid and unit -> (row tile, slice, direction) mapping: fwd_step_kernel's
template <int NB, int MB>
__global__ void __launch_bounds__(256) fwd_step_bf3_kernel(Problem P, StepBuf<float> Sa, StepBuf<float> Sb, int n, int dir_base, int ks, int tiles, int groups,
                                                           int ndirs, int nt) {
    __shared__ float red[4][NB * MB * 8][64];
    const int nbt = (P.B + 31) / 32;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int unit = (within / groups) * 8 + xcd, group = within % groups;
    if (unit >= tiles * ks * ndirs) return;
    const int row_tile = unit % tiles, slice = (unit / tiles) % ks, dir = unit / (tiles * ks);
    if (nt) {
        if (dir + dir_base == 0) fwd_step_bf3<false, NB, MB, true>(P, Sa, n, red, ks, nbt, row_tile, group, slice);
        else fwd_step_bf3<true, NB, MB, true>(P, Sb, n, red, ks, nbt, row_tile, group, slice);
    } else {
        if (dir + dir_base == 0) fwd_step_bf3<false, NB, MB, false>(P, Sa, n, red, ks, nbt, row_tile, group, slice);
        else fwd_step_bf3<true, NB, MB, false>(P, Sb, n, red, ks, nbt, row_tile, group, slice);
    }
}

// The alpha and beta frames of one step share a launch (they are independent chains): twice the workgroups in flight, half the launches.
// fp64: blockIdx.z selects the direction.  fp32 (NB = batch tiles of 32 utterances per workgroup, MB = 16-row blocks per workgroup): a
// one-dimensional grid of (row tile, slice of K, direction) units x `groups` batch tile groups, laid out so that the workgroups of one unit
// -- which stream the SAME matrix tile -- sit on one XCD next to each other (workgroup L goes to XCD L mod 8): L = 8 (unit / 8 x groups +
// group) + unit mod 8.  The last eight units are padded (a workgroup past the end returns).
template <typename R, int NB, int MB, bool HALF>
__global__ void __launch_bounds__(256) fwd_step_kernel(Problem P, StepBuf<R> Sa, StepBuf<R> Sb, int n, int dir_base, int ks, int tiles, int groups, int ndirs, int nt) {
    if constexpr (StepUsesMfma<R>::v) {
        __shared__ float red[4][NB * MB * 8][64];
        const int nbt = (P.B + 31) / 32;
        const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
        const int unit = (within / groups) * 8 + xcd, group = within % groups;
        if (unit >= tiles * ks * ndirs) return;
        const int row_tile = unit % tiles, slice = (unit / tiles) % ks, dir = unit / (tiles * ks);
#ifdef ASG_X_STEP_PROBE
        const long long t_begin = wall_clock64();
#endif
        if (nt) {
            if (dir + dir_base == 0) fwd_step_mfma<false, NB, MB, true, HALF>(P, Sa, n, red, ks, nbt, row_tile, group, slice);
            else fwd_step_mfma<true, NB, MB, true, HALF>(P, Sb, n, red, ks, nbt, row_tile, group, slice);
        } else {
            if (dir + dir_base == 0) fwd_step_mfma<false, NB, MB, false, HALF>(P, Sa, n, red, ks, nbt, row_tile, group, slice);
            else fwd_step_mfma<true, NB, MB, false, HALF>(P, Sb, n, red, ks, nbt, row_tile, group, slice);
        }
#ifdef ASG_X_STEP_PROBE
        {
            // frame 20: every workgroup stamps begin / product done / end (100 MHz ticks); frame 30: one thread prints the summary
            const int wg = blockIdx.x;
            if (n == 20 && threadIdx.x == 0 && wg < 1024) { g_step_probe[wg * 4 + 0] = t_begin; g_step_probe[wg * 4 + 2] = wall_clock64(); }
            if (n == 30 && wg == 0 && threadIdx.x == 0) {
                const int nwg = gridDim.x;
                long long b0 = 0x7fffffffffffffffll, b1 = 0, p0 = b0, p1 = 0, e0 = b0, e1 = 0; int ne = 0; long long esum = 0;
                for (int w = 0; w < nwg && w < 1024; ++w) {
                    const long long b = g_step_probe[w * 4], pd = g_step_probe[w * 4 + 1], e = g_step_probe[w * 4 + 2];
                    if (b == 0) continue;          // (a padding workgroup)
                    b0 = b < b0 ? b : b0; b1 = b > b1 ? b : b1; p0 = pd < p0 ? pd : p0; p1 = pd > p1 ? pd : p1; e0 = e < e0 ? e : e0; e1 = e > e1 ? e : e1;
                    if (e - pd > 100) { ++ne; esum += e - pd; }
                }
                for (int w = 0; w < nwg && w < 1024; ++w)
                    printf("[wg] %d %d %.1f %.1f %.1f\n", w % (int) gridDim.x, w / (int) gridDim.x, (g_step_probe[w * 4] - b0) / 100.0, (g_step_probe[w * 4 + 1] - b0) / 100.0, (g_step_probe[w * 4 + 2] - b0) / 100.0);
                printf("[step probe] %d workgroups: begin %.1f .. %.1f us, product done %.1f .. %.1f us, end %.1f .. %.1f us; %d workgroups ran an epilogue, %.1f us each on average\n",
                       nwg, 0.0, (b1 - b0) / 100.0, (p0 - b0) / 100.0, (p1 - b0) / 100.0, (e0 - b0) / 100.0, (e1 - b0) / 100.0, ne, ne ? esum / 100.0 / ne : 0.0);
            }
        }
#endif
    } else {
        if ((int) blockIdx.z + dir_base == 0) fwd_step_body<R, false>(P, Sa, n);
        else fwd_step_body<R, true>(P, Sb, n);
    }
}


// ------------------------------------------------------------------ full lattice, medium alphabets (64 < N <= 256, fp32)
// One workgroup per chain (utterance x direction), NW = ceil(N / 64) wavefronts, thread i = label i.  The thread keeps
// its normalised transition row (alpha) / column (beta) in REGISTERS (up to 256 floats: one wavefront per SIMD has the
// whole register file) and the frame's exp-domain vector travels through LDS -- the recursion of the small path
// (asg_chains.h) widened over several wavefronts, two workgroup barriers per frame, ALL frames in one launch.  The
// per-frame step launches of fwd_step_kernel are built for N = 10^4 (a 400 MB matrix per frame); at N = 128 every one of
// them is a 10 us round of dependent memory accesses for 4 workgroups of work: 399 launches = 4 ms per cfg-3-sized step.
//   alpha: a_t[i] = x2_t[i] + hmax_i + log2 sum_j Ehat[i][j] p_{t-1}[j],   p = exp2(a - max_i a)      (fully_connected_lattice.cpp:9-29)
//   beta:  y_t = x2_t + b_t,  p = exp2(y - max y),  b_{t-1}[i] = cmax_i + log2 sum_j Fhat[j][i] p[j]  (:32-47)
// Stored states are relative to a per-frame offset (max = 0): the gradient pass (bwd_post_kernel<.., false> + both
// contractions) is offset-free per frame.  A row sum outside [2^-100, 2^100] is redone as an exact log-sum-exp.
// (work_mulog_offset below, for device code: the work area is [emax T B | vectors | maxima | offsets | normaliser log T B])
__host__ __device__ inline size_t mid_au(size_t x) { return (x + 255) & ~(size_t) 255; }
__host__ __device__ inline size_t mid_mulog_offset(size_t elem, int T, int B, int npad) {
    return mid_au((size_t) T * B * elem) + 2 * mid_au(2 * (size_t) B * npad * elem) + 2 * mid_au(3 * (size_t) B * 4) + 2 * mid_au((size_t) B * 8);
}
// (fp32, measured: 256 row elements in one thread spilled into accumulation registers and cost 1360 us at N = 256 where two
// threads per label take 787; 724 -> 676 at N = 192, 575 -> 553 at N = 128.)
// SP threads share a label (SP = 2 for float, 4 for double: at most 128 / 64 row elements = 128 VGPRs per thread) -- thread
// i + q NP holds columns [q NC, (q + 1) NC) of label i's row; parts q >= 1 hand their partial sums over through LDS (one more
// barrier per frame) and otherwise only keep the barriers company.  fp64 (round 4): the reference is double-capable everywhere
// (utils.h:33-39); before, fp64 problems with 64 < N <= 256 took T - 1 launches of fwd_step_kernel<double> (~10 us each).
template <typename R, int NW, int SP>
__global__ void __launch_bounds__(64 * NW * SP) fwd_mid_kernel(Problem P, State W, FwdOut O, int mask) {
    constexpr int NP = 64 * NW;
    constexpr bool SPLIT = true;
    constexpr int NC = NP / SP;                                  // columns of its row a thread holds
    constexpr int NWT = SP * NW;                                 // wavefronts of the workgroup
    __shared__ __attribute__((aligned(16))) R pbuf[NP];          // exp-domain vector of the frame being consumed
    __shared__ R qbuf[2][NP];                                    // its log-domain twin (exact path), double buffered: the exact path
                                                                 // of a slow thread may still read it when a fast one writes the next
    __shared__ R part[SP - 1][NP];                               // partial sums of the other parts
    __shared__ R red[16];
    const int b = blockIdx.x;
    const bool beta = (mask == kFullBeta) || (mask == (kFullAlpha | kFullBeta) && blockIdx.y == 1);
    const int partq = (int) threadIdx.x / NP;                    // which part of the row
    const bool upper = partq > 0;
    const int i = (int) threadIdx.x - partq * NP, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = partq * NC;                                   // first column of this thread's share
    const int N = P.N, T = P.T;
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero(), NINF = Num<R>::ninf();
    const bool rowact = i < N;
    const bool act = rowact && !upper;
    const int ic = rowact ? i : 0;
    const R *tr = (const R *) P.transition;
    R *score_out = (R *) (beta ? O.full_scores : O.full_scores_alpha);
    if (len < 1) {
        if (threadIdx.x == 0 && score_out) score_out[b] = NINF;
        return;
    }
    // this label's row (alpha: Tr[i][.], scores of arriving at i) / column (beta: Tr[.][i], of leaving i), normalised
    const int64_t tbase = beta ? (int64_t) ic * P.ts1 : (int64_t) ic * P.ts0, tstep = beta ? P.ts0 : P.ts1;
    R hmax = NINF;
    for (int j = 0; j < N; ++j) hmax = fmax(hmax, tr[tbase + (int64_t) j * tstep] * L2E);
    hmax = fmax(hmax, LZ);
    V2<R> e2[NC / 2];
#pragma unroll
    for (int j = 0; j < NC / 2; ++j) {
        const int ca = c0 + 2 * j, cb = ca + 1;
        const R ea = (rowact && ca < N) ? Num<R>::exp2(tr[tbase + (int64_t) min(ca, N - 1) * tstep] * L2E - hmax) : R(0);
        const R eb = (rowact && cb < N) ? Num<R>::exp2(tr[tbase + (int64_t) min(cb, N - 1) * tstep] * L2E - hmax) : R(0);
        e2[j] = V2<R>{ea, eb};
    }
    // emissions of this label: frame offset in an SGPR, label offset in a VGPR (32-bit: checked by the launcher)
    __amdgpu_buffer_rsrc_t rin = make_rsrc((R *) P.inputs + (int64_t) b * P.is1, 0xffffffffu);
    const unsigned eoff = (unsigned) (ic * (int) P.is2) * (unsigned) sizeof(R), frame_bytes = (unsigned) P.is0 * (unsigned) sizeof(R);
    auto emis = [&](int f) -> R {
        return buf_load<R>(rin, eoff, (unsigned) __builtin_amdgcn_readfirstlane(gclampi(f, 0, len - 1)) * frame_bytes);
    };
    R *st = (R *) (beta ? W.bh : W.ah) + (int64_t) b * T * N + ic;
    auto wg_max = [&](R v) -> R {              // max over the workgroup (one barrier); every thread gets it
        const R m = wave_allmax(v);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        R r = red[0];
#pragma unroll
        for (int w = 1; w < NWT; ++w) r = fmax(r, red[w]);
        return r;
    };
    auto matvec = [&]() -> R {                 // sum_j e[j] p[j], the vector broadcast from LDS
        V2<R> a0 = {0, 0}, a1 = {0, 0};
#pragma unroll
        for (int q = 0; q < NC / 4; ++q) {
            const V4<R> pv = *reinterpret_cast<const V4<R> *>(&pbuf[c0 + 4 * q]);
            a0 = __builtin_elementwise_fma(e2[2 * q], V2<R>{pv.x, pv.y}, a0);          // v_pk_fma_f32
            a1 = __builtin_elementwise_fma(e2[2 * q + 1], V2<R>{pv.z, pv.w}, a1);
        }
        const V2<R> a = a0 + a1;
        R sum = a.x + a.y;
        if constexpr (SPLIT) {
            // the parts of a row meet in LDS (before the split the 256 row elements of N = 256 spilled into accumulation
            // registers and cost 1007 -> 679 us only by giving up the packed FMAs)
            if (upper) part[partq - 1][i] = sum;
            __syncthreads();
            if (!upper) {
#pragma unroll
                for (int q = 0; q < SP - 1; ++q) sum += part[q][i];
            }
        }
        return sum;
    };
    int qp = 0;                                // qbuf[qp] = the vector being consumed
    auto exact = [&]() -> R {                  // log2 sum_j 2^(Tr2 + q_j) for this label, from the log-domain vector
        R mx = NINF;
        for (int j = 0; j < N; ++j) { const R v = tr[tbase + (int64_t) j * tstep] * L2E + qbuf[qp][j]; mx = (v == v) ? fmax(mx, v) : mx; }
        if (mx == NINF) return NINF;
        R sm = 0;
        for (int j = 0; j < N; ++j) { const R v = tr[tbase + (int64_t) j * tstep] * L2E + qbuf[qp][j]; sm += (v == v) ? Num<R>::exp2(v - mx) : R(0); }
        return mx + Num<R>::log2(sm);
    };
    constexpr int PF = 4;
    double M = 0.0;
    R x2[PF];
    // what the gradient pass needs to recover a frame's row sums from the stored states (bwd_post_kernel<.., true>):
    //   log2 sum_j Ehat[i][j] 2^ah[t-1][j] = ah[t][i] - x2[t][i] - hmax_i + (the normaliser subtracted at frame t)
    // -- the alpha chain logs that normaliser per frame where the streamed step logs its own, and zeros where the streamed
    // step keeps the emissions' frame maxima (none are subtracted here)
    R *ezero = W.work ? (R *) W.work : nullptr;
    R *mulog = W.work ? (R *) ((char *) W.work + mid_mulog_offset(sizeof(R), T, (int) P.B, W.npad)) : nullptr;
    // The frame's normaliser LAGS by one frame: a frame's vector is taken relative to the maximum of the PREVIOUS stored
    // vector, which every wavefront left in red[] before the barrier that published that vector -- so no workgroup-wide
    // maximum (a dependent LDS round trip and a barrier of its own) sits between the product and the next vector.  Stored
    // states are relative to a per-frame offset either way (the gradient pass is offset-free per frame); their maximum is
    // now the growth of one frame instead of 0.
    auto leave_max = [&](R v) { const R m = wave_allmax(v); if (lane == 0) red[wave] = m; };
    auto lagged_max = [&]() -> R {             // (after the barrier that follows leave_max)
        R r = red[0];
#pragma unroll
        for (int w = 1; w < NWT; ++w) r = fmax(r, red[w]);
        return fmax(r, LZ);
    };
    if (!beta) {
        // frame 0
        R a = act ? emis(0) * L2E : NINF;
        R m = fmax(wg_max(a), LZ);
        R ah = a - m;
        M = (double) m;
        if (act) st[0] = ah;
        if (!upper) { pbuf[i] = act ? Num<R>::exp2(ah) : R(0); qbuf[0][i] = act ? ah : NINF; }
        R mu = 0;                                  // max of the stored frame 0: exactly 0
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PF; ++u) x2[u] = emis(1 + u) * L2E;
        for (int t0 = 1; t0 < len; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                if (t < len) {
                    const R xe = x2[u];
                    x2[u] = emis(t + PF) * L2E;
                    const R s = matvec();
                    const R lg = Num<R>::log2(s);
                    R rr = hmax + lg;
                    if (act && !(fabs(lg) < Num<R>::lg_limit())) rr = exact();      // (rare)
                    a = act ? xe + rr : NINF;
                    ah = a - mu;                       // (every thread has read the old vector: the barrier inside matvec)
                    M += (double) mu;
                    if (threadIdx.x == 0 && mulog) { mulog[(int64_t) t * P.B + b] = mu; ezero[(int64_t) t * P.B + b] = R(0); }
                    if (act) st[(int64_t) t * N] = ah;
                    if (!upper) { pbuf[i] = act ? Num<R>::exp2(ah) : R(0); qbuf[qp ^ 1][i] = act ? ah : NINF; }
                    leave_max(ah);
                    __syncthreads();
                    qp ^= 1;
                    mu = lagged_max();
                }
            }
        }
        if (score_out) {
            __syncthreads();
            const R ml = fmax(wg_max(ah), LZ);
            const R sm = wave_allsum(act ? Num<R>::exp2(ah - ml) : R(0));
            __syncthreads();
            if (lane == 0) red[wave] = sm;
            __syncthreads();
            if (threadIdx.x == 0) {
                R tot = red[0];
#pragma unroll
                for (int w = 1; w < NWT; ++w) tot += red[w];
                const double sc = M + (double) ml + (double) Num<R>::log2(tot);
                score_out[b] = (sc < -1e29) ? NINF : (R) (sc * kLn2);
            }
        }
    } else {
        R bh = act ? R(0) : NINF;                     // beta at the last frame
        if (act) st[(int64_t) (len - 1) * N] = R(0);
#pragma unroll
        for (int u = 0; u < PF; ++u) x2[u] = emis(len - 1 - u) * L2E;
        // the first vector is normalised by its own maximum (nothing to lag behind yet)
        R mu = fmax(wg_max(act ? x2[0] + bh : NINF), LZ);
        __syncthreads();
        for (int t0 = len - 1; t0 >= 1; t0 -= PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 - u;
                if (t >= 1) {
                    const R xe = x2[u];
                    x2[u] = emis(t - PF) * L2E;
                    const R y = act ? xe + bh : NINF;
                    const R q = y - mu;
                    M += (double) mu;
                    if (!upper) { pbuf[i] = act ? Num<R>::exp2(q) : R(0); qbuf[qp ^ 1][i] = act ? q : NINF; }
                    leave_max(q);
                    __syncthreads();
                    qp ^= 1;
                    mu = lagged_max();
                    const R s = matvec();              // (its barrier comes after every thread's reads of the vector: the
                    const R lg = Num<R>::log2(s);      //  next frame may rewrite it)
                    R rr = hmax + lg;
                    if (act && !(fabs(lg) < Num<R>::lg_limit())) rr = exact();      // (rare)
                    bh = act ? rr : NINF;
                    if (act) st[(int64_t) (t - 1) * N] = bh;
                }
            }
        }
        if (score_out) {
            // S_full = LSE_i(I_0[i] + beta_0[i])   (fully_connected_lattice.cpp:89)
            const R y = act ? emis(0) * L2E + bh : NINF;
            const R my = fmax(wg_max(y), LZ);
            const R sm = wave_allsum(act ? Num<R>::exp2(y - my) : R(0));
            __syncthreads();
            if (lane == 0) red[wave] = sm;
            __syncthreads();
            if (threadIdx.x == 0) {
                R tot = red[0];
#pragma unroll
                for (int w = 1; w < NWT; ++w) tot += red[w];
                const double sc = M + (double) my + (double) Num<R>::log2(tot);
                score_out[b] = (sc < -1e29) ? NINF : (R) (sc * kLn2);
            }
        }
    }
}

// ------------------------------------------------------------------ full lattice, 256 < N <= 2048 (fp32) / 1024 (fp64): resident slices
// Between the one-workgroup-per-chain kernel above (the row of a label fits its thread's registers up to N = 256) and
// the streamed step (N ~ 10^4: 400 MB per frame) the matrix is 0.25 - 4 MB: too large for one compute unit, far too
// small to be worth a launch per frame (fwd_step_kernel: 14 / 27 us per frame at N = 512 / 1024, a round of dependent
// memory accesses for 14 - 52 workgroups of work).  Here a CLUSTER of G workgroups keeps the whole matrix in registers
// for all frames -- workgroup g the RW = 64 (N <= 512), 32 (N <= 1024) or 16 rows from i0 = g RW, as the A operand of
// v_mfma_f32_16x16x4_f32 (exact fp32, as in the streamed step): of the workgroup's 16 wavefronts, wavefront w has the
// 16-row block w % (RW / 16) and the K part w / (RW / 16) (4, 8 or 16 parts: 8 groups of 16 k = 32 registers per lane; four
// wavefronts per SIMD hide each other's operand reads and bookkeeping: with one per SIMD the product took 6 600 cycles
// for its 128 MFMAs, with four 4 700 for the same 128 per SIMD -- the pipe's rate, 32 cycles apiece whether one
// wavefront issues them or four, tools/ubench/mfma_issue.hip, would be 4 100) -- and
// takes a batch of up to 16 chains of one direction through the frames together:
//   product   s[i][u] = sum_k E[i][k] p_u[k]: the batch's vectors sit in LDS as the B operand ([k / 4][u][k % 4]: one
//             conflict-free ds_read_b128 feeds four MFMAs), 32 MFMAs per wavefront and frame whatever the
//             batch size; the K halves meet in LDS;
//   epilogue  the streamed step's, element for element (same stored state, normaliser log and offsets: the gradient
//             pass and fwd_score_kernel do not know which of the two ran): q = x2 + hmax + log2 s - max of the previous
//             frame; exact log-sum-exp from the stored log-domain state when s leaves [2^-100, 2^100];
//   hand-off  the workgroup's new elements p = 2^q go to the cluster's exchange buffer write-through, in the B-operand
//             layout; once they are acknowledged ONE word says "frame n published"; every workgroup polls its G peers'
//             words and copies the frame's vectors (nb N floats, agent-scope loads) into its LDS, together with
//             the workgroups' maxima of q (the next frame's normaliser).  Double
//             buffered by frame parity: a workgroup can be at most one frame ahead of the slowest reader.
//             (Measured against polling the data itself for a tag in the sign bit: 2.3 vs 5.5 us per frame at N = 512.)
// Clusters are placed with the workgroup index as the slow coordinate (block = g * ncl + c), so a cluster's workgroups
// land on ONE XCD whenever the cluster count is a multiple of 8 and the exchange stays in that XCD's L2.
// fp64 (round 5; the text above describes fp32): the same kernel on v_mfma_f64_16x16x4_f64 with workgroups of 512 threads -- the 32 K
// matrix elements of a workgroup are 128 registers per lane of its 8 wavefronts -- two accumulators, 8-byte exchange words, N <= 1024
// (16 chains x N doubles in LDS).  Per frame at T=400 B=64 N=512 (cycles, ASG_X_CL_PROBE): product 8 500 (128 MFMAs per SIMD at 64
// cycles; only 4 of the 16 chain columns exist at this shape), epilogue 2 700, acknowledge 770, flags 1 500, reload 3 350 = 7 us
// against 11.7 us for a launch per frame (fwd_step_tile_kernel); the step 5.9 -> 4.1 ms, N=1024: 13.1 -> 7.7 ms, N=300: 4.1 -> 3.75.
// All workgroups must be co-resident (they wait for each other): the launcher sizes the grid to the device's compute
// units; a wait that runs out (2^22 polls) poisons the scores with NaN instead of hanging the device.
constexpr int kClNB = 16;                       // chains per batch
// threads per workgroup: fp32 1024 (four wavefronts per SIMD: see above; 32 matrix registers per lane), fp64 512 (two per SIMD: the
// same 32 K elements of the matrix per workgroup are 128 registers per lane, and a lane of a 1024-thread workgroup has 128 in all)
template <typename R> struct ClusterThreads { static constexpr int v = sizeof(R) == 4 ? 1024 : 512; };
constexpr unsigned kClSc1 = 16;   // buffer load aux bit: agent scope
typedef unsigned ClU4 __attribute__((ext_vector_type(4)));
typedef unsigned ClU2 __attribute__((ext_vector_type(2)));
struct ClusterArgs {
    void *xbuf;         // [ncl][2][npadL / 4][kClNB][4]   exp-domain vectors of the frame just produced, in the problem's precision (pad columns stay zero)
    unsigned *xmax;     // [ncl][2][G][kClNB]             key(max q) over each workgroup's rows
    unsigned *flags;    // [ncl][G]                       frames published so far (zero on entry)
    unsigned *fault;    // host-mapped word (or nullptr): incremented when a bounded wait runs out (cluster_fault_word)
    unsigned *callfault;   // device word of THIS call (zero on entry): raised with it -- fwd_repair_kernel, enqueued behind the launch, reads it
    int G, RW, npadL, ncd, cpc, ndirs;
};

// A bounded wait of fwd_cluster_kernel that runs out (part of the grid never became resident: another process on the device, a
// CU-masked stream, a partitioned device) poisons the scores with NaN -- and must not stay silent: the kernel also bumps ONE
// host-pinned word of this process.  The host reads it without a synchronisation: launch_fwd_generic stops taking the resident
// route once it is non-zero (the per-frame launches need no co-residency), asg_cluster_timeouts() reports the count, and
// torch_asg_amd raises on it.  This word is the library's only state between calls; it records faults, it carries no data.
struct ClusterFault {
    unsigned *host = nullptr, *dev = nullptr;
    bool tried = false, warned = false;
};
static ClusterFault &cluster_fault(bool create = true) {
    static ClusterFault F;
    if (!F.tried && create) {
        F.tried = true;
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && h) {
            *(volatile unsigned *) h = 0;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) { F.host = (unsigned *) h; F.dev = (unsigned *) d; }
            else (void) hipHostFree(h);
        }
        (void) hipGetLastError();
    }
    return F;
}

static __host__ __device__ inline size_t cluster_xbuf_floats(int ncl, int npadL) { return (size_t) ncl * 2 * kClNB * npadL; }

extern __shared__ __attribute__((aligned(16))) unsigned char cl_lds_bytes[];
template <typename R>
__global__ void __launch_bounds__(ClusterThreads<R>::v) fwd_cluster_kernel(Problem P, StepBuf<R> Sa, StepBuf<R> Sb, ClusterArgs C, int dir_base) {
    constexpr int kClNT = ClusterThreads<R>::v;
    typedef TileOps<R> Ops;
    R *cl_lds = reinterpret_cast<R *>(cl_lds_bytes);
    __shared__ unsigned pmax[kClNB], lmax[kClNB];
    __shared__ float mus[kClNB];
    __shared__ int lens[kClNB];
    __shared__ double offs[kClNB];
    __shared__ int sfail, smaxlen;
    const int ncl = C.ndirs * C.ncd;
    const int c = (int) blockIdx.x % ncl, g = (int) blockIdx.x / ncl;
    const int dir = dir_base + c / C.ncd, cd = c % C.ncd;
    const bool BETA = dir == 1;
    const StepBuf<R> &S = BETA ? Sb : Sa;
    const int N = P.N, T = P.T, B = P.B, npad = S.npad, RW = C.RW, npadL = C.npadL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    R *pl = cl_lds;                              // [npadL / 4][kClNB][4]: the B operand
    R *red = cl_lds + kClNB * npadL;             // [NKH][RW][kClNB]
    const R L2E = Num<R>::log2e(), LZ = Num<R>::logzero(), NINF = Num<R>::ninf();
    const int MBW = RW / 16, NKH = (kClNT / 64) / MBW;      // row blocks per workgroup, K parts
    const int mb = wave % MBW, kh = wave / MBW;
    const int kspan = npadL / NKH, kbase = kh * kspan, KS4 = kspan / 16;      // this wavefront's K range, in groups of 16
    const int i0 = g * RW;
    // ---- this lane's elements of the (normalised) matrix: row i0 + 16 mb + (lane & 15), k = kbase + 16 s + 4 (lane >> 4) + c
    constexpr int KG = 8192 / kClNT;              // groups of 16 k per wavefront: 128 / (wavefronts per SIMD) matrix registers
    V4<R> ea[KG];
    {
        const int irow = i0 + 16 * mb + (lane & 15);
        const R *src = S.ehat + (int64_t) min(irow, N - 1) * npad;
#pragma unroll
        for (int s4 = 0; s4 < KG; ++s4) {
            const int k = kbase + 16 * s4 + 4 * (lane >> 4);
            V4<R> v = {0, 0, 0, 0};
            if (s4 < KS4 && irow < N && k < npad) v = *reinterpret_cast<const V4<R> *>(src + k);
            ea[s4] = v;
        }
    }
    R *xb = (R *) C.xbuf + (size_t) c * 2 * kClNB * npadL;
    __amdgpu_buffer_rsrc_t rxb = make_rsrc(xb, (unsigned) (2 * kClNB * npadL * sizeof(R)));
    unsigned *fl = C.flags + (size_t) c * C.G;
    unsigned *xm = C.xmax + (size_t) c * 2 * C.G * kClNB;
    const R *tr = (const R *) P.transition;
    if (tid == 0) sfail = 0;
#ifdef ASG_X_CL_PROBE
    unsigned long long pr[5] = {0, 0, 0, 0, 0};
#endif
    unsigned pub = 0;                             // frames this cluster has published (uniform over its workgroups)
#ifdef ASG_X_CL_SPINMAX
    constexpr int kSpinMax = ASG_X_CL_SPINMAX;    // (developer variant: tests/test_hip_variants.py)
#else
    constexpr int kSpinMax = 1 << 22;             // (~0.5 us per poll: a couple of seconds)
#endif
    const int cb0 = cd * C.cpc, cb1 = min(B, cb0 + C.cpc);
    constexpr int IT = 64 * kClNB / kClNT;        // epilogue elements per thread: RW * kClNB <= 64 * 16
    const int n4 = npad / 4;
    for (int rb = cb0; rb < cb1; rb += kClNB) {
        const int nb = min(kClNB, cb1 - rb);
        // epilogue elements: (chain u, row rr_) = (idx & nbm, idx >> nbs), the chain count rounded up to a power of two,
        // so that small batches fill the threads of the first pass instead of a quarter of every pass
        const int nbs = nb <= 1 ? 0 : nb <= 2 ? 1 : nb <= 4 ? 2 : nb <= 8 ? 3 : 4, nbm = (1 << nbs) - 1;
#ifdef ASG_X_CL_NO_FEW
        const bool few = false;                   // (developer A/B)
#else
        const bool few = sizeof(R) == 4 && nb <= 4;      // the product on 4 x 4 blocks (below; fp32 only)
#endif
        R hm[IT];                                 // hmax of the rows this thread finishes
#pragma unroll
        for (int it = 0; it < IT; ++it) hm[it] = S.hmax[min(i0 + ((tid + kClNT * it) >> nbs), N - 1)];
        __syncthreads();
        if (tid < kClNB) {
            const int b = min(rb + tid, B - 1);
            const int len = tid < nb ? (P.in_len ? gclampi(P.in_len[b], 0, T) : T) : 0;
            lens[tid] = len;
            mus[tid] = 0.0f;                      // the first frame's maximum is exactly 0 (fwd_init_kernel)
            offs[tid] = (tid < nb && len >= 1) ? S.off[b] : 0.0;
        }
        if (tid == 0) smaxlen = 0;
        __syncthreads();
        if (tid < nb) atomicMax(&smaxlen, lens[tid]);
        // the vectors of the first frame (fwd_init_kernel wrote them to pbuf[0]), transposed into the operand layout
        for (int idx = tid; idx < kClNB * npadL; idx += kClNT) {
            const int k = idx / kClNB, u = idx - k * kClNB;
            pl[((k >> 2) * kClNB + u) * 4 + (k & 3)] = (u < nb && k < npad) ? S.pbuf[(int64_t) (rb + u) * npad + k] : R(0);
        }
        __syncthreads();
        const int nsteps = smaxlen - 1;
        R xn[IT], wn[IT];                         // emissions (and their frame maxima) of the NEXT frame's elements
        {
            constexpr int NF = 0;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + kClNT * it, u = idx & nbm, rr_ = idx >> nbs, i = i0 + rr_;
                xn[it] = 0; wn[it] = 0;
                if (u < nb && rr_ < RW && i < N) {
                    const int len = lens[u], b = rb + u;
                    if (NF < len - 1) {
                        const int tw = BETA ? len - 2 - (NF) : (NF) + 1;
                        xn[it] = ((const R *) P.inputs)[(int64_t) tw * P.is0 + (int64_t) b * P.is1 + (int64_t) i * P.is2];
                        wn[it] = S.emax[(int64_t) tw * B + b];
                    }
                }
            }
        }
        for (int n = 0; n < nsteps; ++n) {
            const unsigned par = pub & 1u;
#ifdef ASG_X_CL_PROBE
            const unsigned long long c0 = __builtin_readcyclecounter();
#endif
            // (this frame's emissions for the elements this thread finishes were requested during the previous frame's
            // hand-off: no global load is in flight while the matrix pipe runs -- hipcc's waitcnt placement would make the
            // MFMAs wait for it)
            R xe[IT], xw[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) { xe[it] = xn[it]; xw[it] = wn[it]; }
            if (tid < kClNB) { pmax[tid] = fkey(-__builtin_inff()); lmax[tid] = fkey(-__builtin_inff()); }
            // ---- product: 16 rows x 16 chains x this wavefront's K range
            bool few_done = false;
            if constexpr (sizeof(R) == 4) if (few) {
                few_done = true;
                // At most four chains (N = 512 at B = 64: 32 clusters, four chains each): the 16 x 16 x 4 instruction would spend
                // 32 cycles on sixteen chain columns of which four exist.  v_mfma_f32_4x4x1_16B_f32 is sixteen INDEPENDENT 4 x 4
                // outer products in 8 cycles; block (lane >> 2) = (row quad (lane >> 2) & 3, k slot lane >> 4) takes
                // A = E[row 4 quad + (lane & 3)][k] -- which is where this lane's matrix registers already are -- and B = chain
                // (lane & 3)'s element k: the same 256 multiply-adds per instruction, all of them wanted.
                V4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
                const V4<R> *bp = reinterpret_cast<const V4<R> *>(pl) + (size_t) (kbase / 4 + (lane >> 4)) * kClNB + (lane & 3);
                V4<R> b0 = bp[0], b1 = bp[(size_t) min(1, KS4 - 1) * 4 * kClNB];
#pragma unroll
                for (int s4 = 0; s4 < KG; ++s4) {
                    const V4<R> bv = b0;
                    b0 = b1;
                    b1 = bp[(size_t) min(s4 + 2, KS4 - 1) * 4 * kClNB];
                    if (s4 < KS4) {
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ea[s4].x, bv.x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(ea[s4].y, bv.y, acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(ea[s4].z, bv.z, acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(ea[s4].w, bv.w, acc3, 0, 0, 0);
                    }
                }
                const V4f acc = (acc0 + acc1) + (acc2 + acc3);
                // element (row 16 mb + 4 ((lane >> 2) & 3) + q, chain lane & 3) of k slot lane >> 4 sits in register q: the four k
                // slots are the four 16-lane rows of the wavefront -- a reduce-scatter over them (two lane swaps, three adds) leaves
                // the complete sum of register q in row q, one element per lane
                float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3];
                swap_halves(a0, a2);
                swap_halves(a1, a3);
                float xs = a0 + a2, ys = a1 + a3;
                swap_rows(xs, ys);
                red[((size_t) kh * RW + 16 * mb + 4 * ((lane >> 2) & 3) + (lane >> 4)) * 4 + (lane & 3)] = xs + ys;
            }
            if (!few_done) {
                typedef typename Ops::Acc Acc;
                Acc acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};     // (independent chains: the
                                                                                              // matrix pipe never waits for a result)
                const V4<R> *bp = reinterpret_cast<const V4<R> *>(pl) + (size_t) (kbase / 4 + (lane >> 4)) * kClNB + (lane & 15);
                // operand reads two groups ahead of the matrix pipe, unconditional (clamped), so that they are not tied to the
                // trip-count tests around the MFMAs
                V4<R> b0 = bp[0], b1 = bp[(size_t) min(1, KS4 - 1) * 4 * kClNB];
#pragma unroll
                for (int s4 = 0; s4 < KG; ++s4) {
                    const V4<R> bv = b0;
                    b0 = b1;
                    b1 = bp[(size_t) min(s4 + 2, KS4 - 1) * 4 * kClNB];
                    if (s4 < KS4) {
                        acc0 = Ops::mma(ea[s4].x, bv.x, acc0);
                        acc1 = Ops::mma(ea[s4].y, bv.y, acc1);
                        if constexpr (sizeof(R) == 4) {
                            acc2 = Ops::mma(ea[s4].z, bv.z, acc2);
                            acc3 = Ops::mma(ea[s4].w, bv.w, acc3);
                        } else {          // (fp64: two accumulators -- 16 registers that the 128 of the matrix slice do not leave)
                            acc0 = Ops::mma(ea[s4].z, bv.z, acc0);
                            acc1 = Ops::mma(ea[s4].w, bv.w, acc1);
                        }
                    }
                }
                const Acc acc = (acc0 + acc1) + (acc2 + acc3);
                // element (row 16 mb + Ops::row(lane, q), chain lane & 15) sits in register q (fp32: row 4 (lane >> 4) + q, fp64: (lane >> 4) + 4 q)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[((size_t) kh * RW + 16 * mb + Ops::row(lane, q)) * kClNB + (lane & 15)] = acc[q];
            }
            __syncthreads();
#ifdef ASG_X_CL_PROBE
            const unsigned long long c1 = __builtin_readcyclecounter();
#endif
            // ---- epilogue (fwd_step_mfma's, element for element)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + kClNT * it, u = idx & nbm, rr_ = idx >> nbs, i = i0 + rr_;
                if (!(u < nb && rr_ < RW && i < N)) continue;
                const int len = lens[u], b = rb + u;
                if (!(n < len - 1)) continue;
                const int t = BETA ? len - 1 - n : n + 1, tw = BETA ? t - 1 : t;
                R a = 0;
                if (few) { for (int h = 0; h < NKH; ++h) a += red[((size_t) h * RW + rr_) * 4 + u]; }
                else { for (int h = 0; h < NKH; ++h) a += red[((size_t) h * RW + rr_) * kClNB + u]; }
                const R muprev = fmax((R) mus[u], LZ);
                const R lg = Num<R>::log2(a);
                R rr = hm[it] + lg;
                if (!(fabs(lg) < Num<R>::lg_limit())) {
                    // exact rare path: log2-sum-exp2 over j of (Tr2[.][.] + q_j) from the log-domain state (other
                    // workgroups' write-through stores of the previous frame: agent-scope loads)
                    const int tq = BETA ? t : t - 1;
                    const R *stq = S.state + ((int64_t) b * T + tq) * N;
                    const R *inq = (const R *) P.inputs + (int64_t) tq * P.is0 + (int64_t) b * P.is1;
                    const R emq = S.emax[(int64_t) tq * B + b];
                    R mx = NINF;
                    for (int j = 0; j < N; ++j) {
                        const R sj = __hip_atomic_load(&stq[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + sj : sj;
                        const R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                        const R v = trv * L2E + qj;
                        mx = (v == v) ? fmax(mx, v) : mx;
                    }
                    R sm = 0;
                    for (int j = 0; j < N; ++j) {
                        const R sj = __hip_atomic_load(&stq[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const R qj = BETA ? inq[(int64_t) j * P.is2] * L2E - emq + sj : sj;
                        const R trv = BETA ? tr[(int64_t) j * P.ts0 + (int64_t) i * P.ts1] : tr[(int64_t) i * P.ts0 + (int64_t) j * P.ts1];
                        const R v = trv * L2E + qj;
                        sm += (v == v && mx != NINF) ? Num<R>::exp2(v - mx) : R(0);
                    }
                    rr = (mx == NINF) ? mx : mx + Num<R>::log2(sm);
                }
                const R emw = xw[it];
                const R emis = xe[it] * L2E - emw;
                R stv, q;
                if (BETA) { stv = rr - muprev; q = emis + stv; }
                else { stv = emis + rr - muprev; q = stv; }
                __hip_atomic_store(&S.state[((int64_t) b * T + tw) * N + i], stv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                xe[it] = Num<R>::exp2(q);
                atomicMax(&lmax[u], fkey((float) q));
                if (i == 0) {
                    offs[u] += (double) muprev + (double) emw;
                    if (!BETA && S.mulog) S.mulog[(int64_t) tw * B + b] = muprev;
                }
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + kClNT * it, u = idx & nbm, rr_ = idx >> nbs, i = i0 + rr_;
                if (!(u < nb && rr_ < RW && i < N)) continue;
                if (!(n < lens[u] - 1)) continue;
                const unsigned off = (unsigned) (((((size_t) par * (npadL / 4) + (i >> 2)) * kClNB + u) * 4 + (i & 3)) * sizeof(R));
                if constexpr (sizeof(R) == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xe[it]), rxb, off, 0u, kClSc1);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ClU2, xe[it]), rxb, off, 0u, kClSc1);
            }
#ifdef ASG_X_CL_PROBE
            const unsigned long long c2 = __builtin_readcyclecounter();
#endif
            __syncthreads();
            if (tid < nb)
                __hip_atomic_store(&xm[((size_t) par * C.G + g) * kClNB + tid], lmax[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's stores of the frame have been acknowledged
            __syncthreads();
#ifdef ASG_X_CL_PROBE
            const unsigned long long c3 = __builtin_readcyclecounter();
#endif
#ifdef ASG_X_CL_TEST_STALL
            if (tid == 0 && g != 1) __hip_atomic_store(&fl[g], pub + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (workgroup 1 never says so)
#else
            if (tid == 0) __hip_atomic_store(&fl[g], pub + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            {
                const int NF = n + 1;             // the next frame's emissions: in flight under the hand-off (requested any earlier, hipcc's
                                                  // waitcnt placement puts their latency on the epilogue's or the product's path)
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + kClNT * it, u = idx & nbm, rr_ = idx >> nbs, i = i0 + rr_;
                xn[it] = 0; wn[it] = 0;
                if (u < nb && rr_ < RW && i < N) {
                    const int len = lens[u], b = rb + u;
                    if (NF < len - 1) {
                        const int tw = BETA ? len - 2 - (NF) : (NF) + 1;
                        xn[it] = ((const R *) P.inputs)[(int64_t) tw * P.is0 + (int64_t) b * P.is1 + (int64_t) i * P.is2];
                        wn[it] = S.emax[(int64_t) tw * B + b];
                    }
                }
            }
            }
            // ---- wait for the G workgroups of the cluster, then take the frame
            if (tid < C.G) {
                int spins = 0;
                while (__hip_atomic_load(&fl[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pub + 1u && spins < kSpinMax) ++spins;
                if (spins >= kSpinMax) sfail = 1;
            }
            __syncthreads();
#ifdef ASG_X_CL_PROBE
            const unsigned long long c4 = __builtin_readcyclecounter();
#endif
            if (sfail) break;
            {
                // (all of a thread's loads go out before the first of them is used: one memory latency, not sixteen)
                // (in 16-byte pieces: GR = 1 (fp32) or 2 (fp64) per chain and group of four k)
                constexpr int GR = sizeof(R) / 4;
                const int total = n4 * nb * GR;
                // (fp32, 8 / 16: +4 % at N = 1024, -7 / -15 % at N = 512 -- the matrix rows spill into AGPRs; fp64, 8: the reload 10 300 -> 9 450
                // cycles at N = 1024, 3 350 -> 4 000 at N = 512; 16: spills, 7 100 / 9 800)
                constexpr int RL = 4;
                // fp64: the workgroups' maxima of q are requested FIRST and used last, their latency passes under the vectors' (step at
                // T=400 B=64: N = 512 3.81 -> 3.72 ms, N = 1024 7.68 -> 7.46).  fp32 keeps them as a loop of its own after the copy: with
                // 128 registers per lane the two more cost more than the round trip (N = 512 1.89 -> 1.94 ms, N = 1024 3.24 -> 3.28).
                constexpr bool kMaxFirst = sizeof(R) == 8;
                constexpr int XM = kMaxFirst ? 128 * kClNB / kClNT : 1;       // (G <= 128)
                unsigned xk[XM];
                if constexpr (kMaxFirst) {
#pragma unroll
                    for (int j = 0; j < XM; ++j) {
                        const int idx = tid + kClNT * j, u = idx & (kClNB - 1);
                        xk[j] = 0u;                           // (below every key)
                        if (idx < C.G * kClNB && u < nb && n < lens[u] - 1)
                            xk[j] = __hip_atomic_load(&xm[(size_t) par * C.G * kClNB + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                for (int base = 0; base < total; base += kClNT * RL) {
                    ClU4 v[RL];
                    int dst[RL];
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int idx = base + tid + kClNT * k, ic = min(idx, total - 1);
                        const int piece = ic % GR, cu = ic / GR, k4 = cu / nb, u = cu - k4 * nb;
                        dst[k] = (idx < total && n < lens[u] - 1) ? (k4 * kClNB + u) * (int) (4 * sizeof(R)) + 16 * piece : -1;       // (bytes)
                        const unsigned off = (unsigned) ((((size_t) par * (npadL / 4) + k4) * kClNB + u) * (4 * sizeof(R)) + 16 * piece);
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rxb, off, 0u, kClSc1);      // (agent scope: the line may sit, two frames old, in this XCD's L2)
                    }
#pragma unroll
                    for (int k = 0; k < RL; ++k)
                        if (dst[k] >= 0) *reinterpret_cast<ClU4 *>(reinterpret_cast<unsigned char *>(pl) + dst[k]) = v[k];
                }
                // the frame's normaliser: max q over the cluster's workgroups (keys: max is order-independent)
                if constexpr (kMaxFirst) {
#pragma unroll
                    for (int j = 0; j < XM; ++j)
                        if (xk[j] != 0u) atomicMax(&pmax[(tid + kClNT * j) & (kClNB - 1)], xk[j]);
                } else {
                    // (handing this loop to the LAST threads, which have no part of the copy at N = 512, measured slower: 1.90 -> 2.00 ms)
                    for (int idx = tid; idx < C.G * kClNB; idx += kClNT) {
                        const int u = idx & (kClNB - 1);
                        if (u < nb && n < lens[u] - 1)
                            atomicMax(&pmax[u], __hip_atomic_load(&xm[(size_t) par * C.G * kClNB + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    }
                }
            }
            __syncthreads();
            if (sfail) break;
            if (tid < nb && n < lens[tid] - 1) mus[tid] = funkey(pmax[tid]);
            __syncthreads();
#ifdef ASG_X_CL_PROBE
            const unsigned long long c5 = __builtin_readcyclecounter();
            if (blockIdx.x == 0 && tid == 0) { pr[0] += c1 - c0; pr[1] += c2 - c1; pr[2] += c3 - c2; pr[3] += c4 - c3; pr[4] += c5 - c4; }
#endif
            ++pub;
        }
        if (sfail) break;
        if (g == 0 && tid < nb && lens[tid] >= 1) S.off[rb + tid] = offs[tid];
    }
#ifdef ASG_X_CL_PROBE
    if (blockIdx.x == 0 && tid == 0)
        printf("[cluster probe] frames %u: product %llu  epilogue %llu  ack+barrier %llu  flags %llu  reload %llu cycles per frame (s_memtime ticks)\n",
               pub, pr[0] / max(pub, 1u), pr[1] / max(pub, 1u), pr[2] / max(pub, 1u), pr[3] / max(pub, 1u), pr[4] / max(pub, 1u));
#endif
    if (sfail && g == 0 && tid < kClNB)
        for (int b = cb0 + tid; b < cb1; b += kClNB) S.off[b] = __builtin_nan("");      // (never hang, never return a wrong number quietly)
    if (sfail && tid == 0) {
        if (C.callfault) __hip_atomic_store(C.callfault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (C.fault) __hip_atomic_fetch_add(C.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the resident-slice route: 256 < N <= 2048 (fp32) / 1024 (fp64: the batch's vectors, 16 chains x N, have to fit the LDS beside the
// K parts' partial sums -- 144 KB at N = 1024); ASG_NO_CLUSTER=1: the per-frame launches instead
static constexpr int cluster_max_n(size_t elem) { return elem == 4 ? 2048 : 1024; }
static bool cluster_alphabet(const Problem &P, size_t elem) {
    if (P.N <= 256 || P.N > cluster_max_n(elem)) return false;
    return !(knobs().no_cluster > 0);
}
constexpr size_t kClusterBytes = 8u << 20;       // exchange vectors of every cluster
#ifndef ASG_X_TILE_MAX_N32
#define ASG_X_TILE_MAX_N32 0
#endif
// fp32: the 16 x 16 tile step up to this alphabet.  0 = never: measured at T = 400, B = 64 (developer builds with 3072) it LOSES to
// fwd_step_mfma's pre-tiled operands and 80-row tiles -- N = 1500 17.7 against 12.6 ms per step, N = 2048 24.3 against 15.3, N = 3000
// 54 against 24 -- so in fp32 the tile step stays a developer switch and fp64 is what it is for.
constexpr int kTileStepMaxN32 = ASG_X_TILE_MAX_N32;

// the medium-alphabet route: fp32, 64 < N <= 256, 32-bit emission offsets (ASG_NO_MID=1: the per-frame launches instead)
static bool mid_alphabet(const Problem &P, size_t elem) {
    if (P.N <= 64 || P.N > 256) return false;
    if (knobs().no_mid > 0) return false;
    const double fr = (double) (P.T - 1) * (double) P.is0 * (double) elem, ln = (double) (P.N - 1) * (double) P.is2 * (double) elem;
    return P.is0 >= 0 && P.is2 >= 0 && fr < 4294967296.0 && ln < 2147483648.0;
}

// scores: grid = B, block = 256.  alpha: A + LSE_i(ah[len-1]);  beta: C_0 + LSE_i(q_0), q_0 = I2[0]-emax[0]+bh[0]
template <typename R, bool BETA>
__global__ void __launch_bounds__(256) fwd_score_kernel(Problem P, StepBuf<R> S, R *scores) {
    __shared__ R red[4];
    const int b = blockIdx.x, N = P.N, T = P.T;
    const int len = P.in_len ? gclampi(P.in_len[b], 0, T) : T;
    if (len < 1) { if (threadIdx.x == 0) scores[b] = Num<R>::ninf(); return; }
    const int t = BETA ? 0 : len - 1;
    const R *st = S.state + ((int64_t) b * T + t) * N;
    const R *in = (const R *) P.inputs + (int64_t) t * P.is0 + (int64_t) b * P.is1;
    const R em = S.emax[(int64_t) t * P.B + b];
    R m = Num<R>::ninf();
    for (int i = threadIdx.x; i < N; i += 256) {
        R q = BETA ? in[(int64_t) i * P.is2] * Num<R>::log2e() - em + st[i] : st[i];
        m = fmax(m, q);
    }
    m = fmax(block_reduce_max<R>(m, red), Num<R>::logzero());
    R s = 0;
    for (int i = threadIdx.x; i < N; i += 256) {
        R q = BETA ? in[(int64_t) i * P.is2] * Num<R>::log2e() - em + st[i] : st[i];
        s += Num<R>::exp2(q - m);
    }
    s = block_reduce_sum<R>(s, red);
    if (threadIdx.x == 0) {
        double sc = S.off[b] + (double) m + (double) Num<R>::log2(s);
        scores[b] = (sc < -1e29) ? Num<R>::ninf() : (R) (sc * kLn2);
    }
}


}  // namespace

unsigned cluster_timeouts() {
    ClusterFault &F = cluster_fault(false);         // (a query never allocates: nothing can have timed out before the first launch)
    return F.host ? *(volatile unsigned *) F.host : 0u;
}

// ---------------------------------------------------------------------------------------------- host side
static int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void) hipGetLastError();
        cus = 256;
    }
    return cus;
}
// Shape of the fp32 streaming step's grid: MB 16-row blocks per workgroup (kStepMBMin .. kStepMB) and ks slices of K (1 .. kStepMaxSlices, at least
// ~8 chunks of 32 k each), chosen by a cost model fitted to measurements (tools/step_mb_time.sh, tools/step_ks_time.sh; T=400, B = 32 .. 128,
// N = 1500 .. 7000): a frame costs
//     rounds x ((MB x 48 ns + 20 ns) x chunks per slice x batch tiles per workgroup  +  (ks - 1) x 4.5 us),   rounds = ceil(workgroups / compute units)
// -- the product of one workgroup over its rows, its read of the batch's vectors (whatever its height), and the partial-sum exchange of
// its slices (write-through stores, a ticket, the last arriver's reads).
// It orders every measured pair correctly: N = 3000 at B = 64 takes 48-row tiles (252 workgroups) instead of 80-row ones (152 on 256
// compute units): 51.4 -> 42.4 us per frame; N = 2100: 48-row tiles WITHOUT slices beat 80-row tiles with two (33.0 against 36.9);
// N = 3500 / 4000: 64-row tiles; N = 5000 and cfg 5: 80.
constexpr int kStepMaxSlices = 8, kStepMBMin = 2;
static double step_cost(int N, int groups, int dirs, int nb, int mb, int ks, int cus) {
    const int nchunks = ((N + 3) / 4 * 4 + 31) / 32;
    const long wgs = (long) ((N + 16 * mb - 1) / (16 * mb)) * groups * dirs * ks, rounds = (wgs + cus - 1) / cus;
    return (double) rounds * (((double) mb * 0.048 + 0.02) * ((nchunks + ks - 1) / ks) * (nb == 2 ? 1.9 : 1.0) + (ks - 1) * 4.5);
}
static int step_slices(int N, int groups, int dirs, int nb, int mb, int cus) {
    const int nchunks = ((N + 3) / 4 * 4 + 31) / 32;
    int best = 1;
    double best_cost = step_cost(N, groups, dirs, nb, mb, 1, cus);
    for (int ks = 2; ks <= kStepMaxSlices && ks <= nchunks / 8; ++ks) {
        const double c = step_cost(N, groups, dirs, nb, mb, ks, cus);
        if (c < best_cost) { best = ks; best_cost = c; }
    }
    return best;
}
// Batch tiles of 32 utterances per workgroup (1 or 2) and tile height, priced together.  Two batch tiles per workgroup read the matrix
// once for both (priced at 1.9 products instead of 2) but halve the workgroups; one per workgroup leaves the sharing to sibling workgroups on
// one XCD.  Measured (T=400, ms per step, two / one): B = 96 N = 2500 24.5 / 17.5 (three batch tiles: the second group is half empty),
// B = 128 N = 1500 12.6 / 10.3, B = 128 N = 3000 29.1 / 30.6, B = 256 N = 3000 (T=200) 32.8 / 36.7; at B = 64 one always (N = 3000: 21.7 / 15.8).
// A function of the problem's shape only: the operand-order copies of the matrix are laid out for the height before the recursion knows
// which directions it runs (priced for both; the evaluation route runs one and re-prices its slices).  ASG_STEP_ONE_TILE=1/0 and
// ASG_STEP_ROW_BLOCKS force either.
struct StepPlan { int nb, mb; };
static StepPlan step_plan(int N, int B, int cus) {
    const int nbt = (B + 31) / 32;
    int nb_lo = 1, nb_hi = nbt > 1 ? 2 : 1;
    if (knobs().step_one_tile == 1) nb_hi = 1;
    else if (knobs().step_one_tile == 0 && nbt > 1) nb_lo = 2;
    int mb_lo = kStepMBMin, mb_hi = kStepMB;
    if (knobs().step_row_blocks >= kStepMBMin && knobs().step_row_blocks <= kStepMB) mb_lo = mb_hi = knobs().step_row_blocks;
    StepPlan best{nb_lo, mb_hi};
    double best_cost = -1;
    for (int nb = nb_lo; nb <= nb_hi; ++nb) {
        const int groups = (nbt + nb - 1) / nb;
        for (int mb = mb_hi; mb >= mb_lo; --mb) {
            const double c = step_cost(N, groups, 2, nb, mb, step_slices(N, groups, 2, nb, mb, cus), cus);
            if (best_cost < 0 || c < best_cost) { best = StepPlan{nb, mb}; best_cost = c; }
        }
    }
    return best;
}
// The streaming step on the bfloat16 pipe (fwd_step_bf3): fp32 problems of more than 64 utterances (three batch tiles of 32 or more).  Up to
// 64 the frame is bound by what a compute unit can load (its matrix tile + the batch tile's vectors), and three planes are 1.5x the bytes
// of a float: measured T=400, bf3 / fp32 ms per step, B = 48 N = 2048 8.9 / 7.8, B = 64 N = 1500 7.5 / 7.3, N = 3000 14.4 / 14.2, N = 5000
// 38.0 / 31.4.  From three batch tiles on the matrix instructions take over and the bfloat16 pipe wins: B = 96 N = 3000 17.7 / 20.4,
// B = 128 N = 2048 12.5 / 13.2, N = 3000 21.7 / 26.0, N = 5000 52.2 / 59.1, B = 192 N = 2500 (T=200) 12.4 / 15.3, B = 256 N = 1500 8.0 / 9.0,
// N = 3000 19.1 / 29.9 (tools/shape_times.py, ASG_STEP_NO_BF3=1 for the fp32 instruction; profiles/r06_step_bf3_ab.txt).
// A function of the shape (the buffers are sized by it); ASG_STEP_NO_BF3=1 keeps the fp32 instruction (tests, A/B).
static bool step_bf3_shape(int elem, int B) { return elem == 4 && StepUsesMfma<float>::v && B > (knobs().step_bf3_min_b > 0 ? knobs().step_bf3_min_b - 1 : 64); }
static bool step_bf3(int elem, int B) { return step_bf3_shape(elem, B) && !(knobs().step_no_bf3 > 0); }

template <typename R>
hipError_t launch_prep_generic(const Problem &P, const State &W, hipStream_t stream) {
    hipLaunchKernelGGL((prep_kernel<R, false>), dim3(P.N), dim3(256), 0, stream, (const R *) P.transition, P.ts0, P.ts1,
                       P.N, W.npad, (R *) W.ehat, (R *) W.rmax);
    // medium alphabets: fwd_mid_kernel normalises its rows / columns itself and the gradient pass reads ehat / rmax only --
    // no column-normalised twin, no operand-order copies (40 us of a 580 us step at T=400 B=64 N=128)
    if (mid_alphabet(P, sizeof(R))) return hipGetLastError();
    {
        // column-normalised twin: column maxima first.  Their [N] 64-bit keys borrow the head of the forward work area,
        // which nothing uses before the recursion's own set-up kernels run (later on this stream); it is >= 16 B npad bytes
        unsigned long long *keys = (unsigned long long *) W.work;
        if (!keys) return hipErrorInvalidValue;
        hipError_t me = zero_async(keys, (size_t) P.N * sizeof(unsigned long long), stream);      // 0 < key(-inf)
        if (me != hipSuccess) return me;
        hipLaunchKernelGGL((colmax_kernel<R>), dim3((P.N + 63) / 64, (P.N + 255) / 256), dim3(256), 0, stream,
                           (const R *) P.transition, P.ts0, P.ts1, P.N, keys);
        hipLaunchKernelGGL((colnorm_kernel<R>), dim3((W.npad + 63) / 64, (P.N + 63) / 64), dim3(256), 0, stream,
                           (const R *) P.transition, P.ts0, P.ts1, P.N, W.npad, keys, (R *) W.fhat, (R *) W.cmax);
    }
    if constexpr (StepUsesMfma<R>::v) {
        if (!W.etile || !W.ftile) return hipErrorInvalidValue;
        const int mb = step_plan(P.N, P.B, device_cus()).mb;
        if (step_bf3((int) sizeof(R), P.B)) {
            hipLaunchKernelGGL(tile3_kernel, dim3(4096), dim3(256), 0, stream, (const float *) W.ehat, P.N, W.npad, mb, (U4v *) W.etile);
            hipLaunchKernelGGL(tile3_kernel, dim3(4096), dim3(256), 0, stream, (const float *) W.fhat, P.N, W.npad, mb, (U4v *) W.ftile);
        } else {
            hipLaunchKernelGGL(tile_kernel, dim3(4096), dim3(256), 0, stream, (const float *) W.ehat, P.N, W.npad, mb, (float *) W.etile);
            hipLaunchKernelGGL(tile_kernel, dim3(4096), dim3(256), 0, stream, (const float *) W.fhat, P.N, W.npad, mb, (float *) W.ftile);
        }
    }
    return hipGetLastError();
}

size_t step_tile_bytes_generic(int elem, int N, int B) {
    if (!(elem == 4 && StepUsesMfma<float>::v)) return 0;
    const size_t f32 = step_tile_floats_max(N) * sizeof(float), b3 = step_bf3_shape(elem, B) ? step_tile3_units_max(N) * 16 : 0;
    return f32 > b3 ? f32 : b3;
}

// the vectors of the fp32 streaming step a second time, in operand order (step_ptile_index): two frames per direction, behind the
// normaliser log
static size_t step_ptile_bytes(int elem, int B, int N) {
    if (!(elem == 4 && StepUsesMfma<float>::v)) return 0;
    const size_t f32 = au(2 * step_ptile_floats(B, (N + 3) / 4 * 4) * sizeof(float));
    const size_t b3 = step_bf3_shape(elem, B) ? au(2 * step_ptile3_elems(B, (N + 3) / 4 * 4) * sizeof(unsigned short)) : 0;
    return f32 > b3 ? f32 : b3;
}
// tickets and partial sums of the K slices, sized for any tile height: one ticket per (row tile, batch tile) -- most at 3 row blocks --
// and 2 MB x 256 floats per slice and tile, row tiles x MB <= N / 16 + kStepMB
static size_t step_ticket_bytes(int B, int N) { return au((size_t) ((N + 31) / 32) * ((B + 31) / 32) * sizeof(unsigned)); }
static size_t step_partial_bytes(int B, int N) {
    return au((size_t) ((N + 15) / 16 + kStepMB) * ((B + 31) / 32) * kStepMaxSlices * 2 * 256 * sizeof(float));
}
// forward work buffers live behind the saved state (see fwd_work_bytes_generic): emax, pbuf x2 dirs, mu, off
size_t fwd_work_bytes_generic(int elem, int T, int B, int N) {
    const size_t npad = (size_t) (N + 3) / 4 * 4;
    return au((size_t) T * B * elem) + 2 * au(2 * (size_t) B * npad * elem) + 2 * au(3 * (size_t) B * 4) + 2 * au((size_t) B * 8) +
           au((size_t) T * B * elem) + 2 * step_ptile_bytes(elem, B, N) +
           (step_ptile_bytes(elem, B, N) ? 2 * (step_ticket_bytes(B, N) + step_partial_bytes(B, N)) : 0) + ((N > 256 && N <= cluster_max_n((size_t) elem)) ? kClusterBytes : 0) + 4096;
}
template <typename R>
hipError_t launch_fwd_full_generic(const Problem &P, const State &W, const FwdOut &O, int full_mask, bool store, hipStream_t stream) {
    if (full_mask && mid_alphabet(P, sizeof(R))) {
        dim3 grid(P.B, __builtin_popcount(full_mask));
        const int nw = (P.N + 63) / 64;
        constexpr int SP = sizeof(R) == 4 ? 2 : 4;          // threads per label (fwd_mid_kernel)
        if (nw <= 2) hipLaunchKernelGGL((fwd_mid_kernel<R, 2, SP>), grid, dim3(128 * SP), 0, stream, P, W, O, full_mask);
        else if (nw == 3) hipLaunchKernelGGL((fwd_mid_kernel<R, 3, SP>), grid, dim3(192 * SP), 0, stream, P, W, O, full_mask);
        else hipLaunchKernelGGL((fwd_mid_kernel<R, 4, SP>), grid, dim3(256 * SP), 0, stream, P, W, O, full_mask);
    } else if (full_mask) {
        if (!W.work) return hipErrorInvalidValue;
        char *wk = (char *) W.work;
        const size_t e = sizeof(R);
        R *emax = (R *) wk; wk += au((size_t) P.T * P.B * e);
        hipLaunchKernelGGL((emax_kernel<R>), dim3(P.T, P.B), dim3(256), 0, stream, P, emax);
        // the p vectors are npad wide: their pad columns must be (and stay) zero
        // (... and the arrival counters of the one-launch route, at the very end of the work area)
        char *bar_area = (char *) W.work + fwd_work_bytes_generic((int) e, P.T, P.B, P.N) - 4096;
        // (per direction: a caller may run the two directions as two calls on two streams -- each clears only what is its own)
        const bool want[2] = {(full_mask & kFullAlpha) != 0, (full_mask & kFullBeta) != 0};
        const size_t dirblock = au(2 * (size_t) P.B * W.npad * e) + au(3 * (size_t) P.B * 4) + au((size_t) P.B * 8);
        for (int dir = 0; dir < 2; ++dir)
            if (want[dir]) (void) zero_async(wk + dir * dirblock, dirblock, stream);
        (void) bar_area;
        StepBuf<R> Sd[2];
        for (int dir = 0; dir < 2; ++dir) {
            StepBuf<R> S{};
            S.pbuf = (R *) wk; wk += au(2 * (size_t) P.B * W.npad * e);
            S.mu = (unsigned *) wk; wk += au(3 * (size_t) P.B * 4);
            S.off = (double *) wk; wk += au((size_t) P.B * 8);
            S.emax = emax;
            S.npad = W.npad;
            const bool beta = dir == 1;
            S.state = (R *) (beta ? W.bh : W.ah);
            S.ehat = (const R *) (beta ? W.fhat : W.ehat);
            S.etile = (const R *) (beta ? W.ftile : W.etile);
            S.hmax = (const R *) (beta ? W.cmax : W.rmax);
            S.mulog = beta ? nullptr : (R *) ((char *) W.work + work_mulog_offset(e, P.T, P.B, W.npad));
            S.bf3 = step_bf3((int) e, P.B) ? 1 : 0;
            if (const size_t pbytes = step_ptile_bytes((int) e, P.B, P.N)) {
                char *area = (char *) W.work + work_mulog_offset(e, P.T, P.B, W.npad) + au((size_t) P.T * P.B * e);
                S.ptile = (R *) (area + dir * pbytes);
                if (want[dir]) (void) zero_async(S.ptile, pbytes, stream);       // (pad positions stay zero)
                const size_t tb = step_ticket_bytes(P.B, P.N), sb = step_partial_bytes(P.B, P.N);
                S.tickets = (unsigned *) (area + 2 * pbytes + dir * tb);
                if (want[dir]) (void) zero_async(S.tickets, tb, stream);
                S.partial = (R *) (area + 2 * pbytes + 2 * tb + dir * sb);
            }
            Sd[dir] = S;
        }
        const bool do_a = full_mask & kFullAlpha, do_b = full_mask & kFullBeta;
        if (do_a) hipLaunchKernelGGL((fwd_init_kernel<R, false>), dim3(P.B), dim3(256), 0, stream, P, Sd[0]);
        if (do_b) hipLaunchKernelGGL((fwd_init_kernel<R, true>), dim3(P.B), dim3(256), 0, stream, P, Sd[1]);
        const int srows = StepUsesMfma<R>::v ? 16 * kStepMB : 64;
        dim3 sgrid((P.N + srows - 1) / srows, (P.B + 31) / 32, (do_a && do_b) ? 2 : 1);
        bool stepped = false;
        {
            if (cluster_alphabet(P, e) && P.T >= 2) {
                constexpr int kClNT = ClusterThreads<R>::v;
                int dev = 0, cus = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                    cus = 256;
                // (a CU-masked stream sees fewer compute units than the device has: the grid is sized to what the STREAM may use)
                {
                    uint32_t mask[16] = {0};
                    if (hipExtStreamGetCUMask(stream, 16, mask) == hipSuccess) {
                        int bits = 0;
                        for (int q = 0; q < 16; ++q) bits += __builtin_popcount(mask[q]);
                        if (bits > 0 && bits < cus) cus = bits;
                    } else {
                        (void) hipGetLastError();
                    }
                }
                ClusterArgs C{};
                // (fp64 up to 512 labels: 32 rows per workgroup, not 64 -- at T=400 B=64 N=512 the product 8 500 -> 4 600 cycles per frame for
                // +2 000 of epilogue, flags and reload with twice the peers: 16 600 -> 14 400 in all; 16 rows: 15 500)
                C.RW = P.N <= 512 ? (sizeof(R) == 8 ? 32 : 64) : P.N <= 1024 ? 32 : 16;
#ifdef ASG_DEV_PROBES
                // (a workgroup's registers hold 32 K matrix elements: more rows than that allows would silently drop part of K)
                if (const char *ev = getenv("ASG_CL_RW"))
                    if ((atoi(ev) == 16 || atoi(ev) == 32 || atoi(ev) == 64) && (size_t) atoi(ev) * ((W.npad + 255) / 256 * 256) <= 32768) C.RW = atoi(ev);
#endif
                C.G = (P.N + C.RW - 1) / C.RW;
                C.npadL = (W.npad + 255) / 256 * 256;
                C.ndirs = (do_a && do_b) ? 2 : 1;
                int ncd = cus / C.G / C.ndirs;
                if (ncd < 1) ncd = 1;
                if (ncd > P.B) ncd = P.B;
                C.cpc = (P.B + ncd - 1) / ncd;
                C.ncd = (P.B + C.cpc - 1) / C.cpc;
                const int ncl = C.ndirs * C.ncd;
                char *ca = bar_area - kClusterBytes;
                const size_t xb = au(cluster_xbuf_floats(ncl, C.npadL) * sizeof(R)), xmb = au((size_t) ncl * 2 * C.G * kClNB * 4), flb = au((size_t) ncl * C.G * 4);
                // beyond 1024 labels a cluster is half the device (one per direction) and takes the batch 16 chains at a time: worth it
                // for ONE round (B <= 16).  (Round 4: up to three; since the streaming step picks its tile height the launch per frame wins
                // from two rounds on -- ms per step at T=400, cluster / launch per frame: N = 1500 B = 16 3.65 / 5.60, B = 32 6.97 / 6.32,
                // B = 48 10.5 / 7.6; N = 2048 B = 16 4.15 / 6.95, B = 32 7.93 / 8.02, B = 48 11.96 / 8.72.)
                const bool few_rounds = P.N <= 1024 || (C.cpc + kClNB - 1) / kClNB <= 1;
                // (at least 84 KB of LDS: a compute unit then holds ONE of these workgroups -- two on one unit would share its
                // matrix pipes and make their whole clusters wait, while other units stay empty)
                size_t lds = ((size_t) kClNB * C.npadL + (size_t) (kClNT / 64) * 16 * kClNB) * sizeof(R);
                if (lds < 84 * 1024) lds = 84 * 1024;
                // The workgroups of a cluster wait for each other: the route is taken only if (a) no earlier launch of this process
                // ever timed out (cluster_fault), (b) the kernel can have its LDS (per device and cheap: asked on every call) and the
                // runtime says a workgroup of it fits a compute unit, (c) the grid fits the compute units this stream may use (above).
                // Otherwise: the per-frame launches below, which need none of it.
                ClusterFault &CF = cluster_fault();
                bool resident_ok = CF.host != nullptr && *(volatile unsigned *) CF.host == 0u;
                if (CF.host && !resident_ok && !CF.warned) {
                    CF.warned = true;
                    fprintf(stderr, "[torch_asg_amd] a resident-slice forward launch timed out earlier in this process (its scores were NaN): "
                                    "taking the per-frame launches from now on\n");
                }
                if (resident_ok && hipFuncSetAttribute((const void *) fwd_cluster_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) {
                    (void) hipGetLastError();
                    resident_ok = false;
                }
                if (resident_ok) {
                    int per_cu = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *) fwd_cluster_kernel<R>, kClNT, lds) != hipSuccess || per_cu < 1) {
                        (void) hipGetLastError();
                        resident_ok = false;
                    }
                }
                if (resident_ok && few_rounds && ncl * C.G <= cus && xb + xmb + flb + 256 <= kClusterBytes) {
                    C.xbuf = ca;
                    C.xmax = (unsigned *) (ca + xb);
                    C.flags = (unsigned *) (ca + xb + xmb);
                    C.callfault = (unsigned *) (ca + xb + xmb + flb);
                    C.fault = CF.dev;
                    (void) zero_async(ca, xb + xmb + flb + 256, stream);
                    hipLaunchKernelGGL((fwd_cluster_kernel<R>), dim3(ncl * C.G), dim3(kClNT), lds, stream, P, Sd[0], Sd[1], C, do_a ? 0 : 1);
                    // ... and, behind it, the repair of THIS call should one of its waits have run out (returns at once otherwise)
                    hipLaunchKernelGGL((fwd_repair_kernel<R>), dim3((P.B + 15) / 16, C.ndirs), dim3(256), 0, stream, P, Sd[0], Sd[1],
                                       (const unsigned *) C.callfault, do_a ? 0 : 1);
                    stepped = true;
                }
            }
        }
        {
            // below the streaming regime (and, fp32, where the resident-slice kernel did not take the problem): 16 x 16 tiles on the
            // matrix instruction of the problem's precision (ASG_NO_TILE_STEP=1: the kernels built for N = 10^4)
            const int tile_max_n = sizeof(R) == 8 ? 2048 : kTileStepMaxN32;
            if (!stepped && P.N <= tile_max_n && !(knobs().no_tile_step > 0)) {
                const int nbt = (P.N > 512 && P.B > 16) ? 2 : 1;
                const dim3 dgrid((P.N + 15) / 16, (P.B + 16 * nbt - 1) / (16 * nbt), (do_a && do_b) ? 2 : 1);
                for (int n = 0; n + 1 < P.T; ++n) {
                    if (nbt == 2) hipLaunchKernelGGL((fwd_step_tile_kernel<R, 2>), dgrid, dim3(256), 0, stream, P, Sd[0], Sd[1], n, do_a ? 0 : 1);
                    else hipLaunchKernelGGL((fwd_step_tile_kernel<R, 1>), dgrid, dim3(256), 0, stream, P, Sd[0], Sd[1], n, do_a ? 0 : 1);
                }
                stepped = true;
            }
        }
        if (!stepped) {
            // fp32: row tiles of 48 / 64 / 80 rows and K split over ks workgroups per (row tile, batch tile), whichever grid the cost model
            // prices lowest (step_cost; cfg 5: 125 row tiles of 80 rows x 2 directions, no slices = 250 workgroups).
            // One or two batch tiles of 32 utterances per workgroup: step_plan.
            int ks = 1, nb = 1, mb = kStepMB, tiles = 0, groups = 1, ndirs = 1, nt = 1;
            const bool half = P.B <= 16 && !(knobs().step_full_tile > 0);          // (at most 16 utterances: half of the 32-utterance tile)
            if constexpr (StepUsesMfma<R>::v) {
                const int cus = device_cus();
                const StepPlan plan = step_plan(P.N, P.B, cus);
                nb = plan.nb;
                mb = plan.mb;          // (the height launch_prep_generic laid the operand-order copies out for)
                tiles = (P.N + 16 * mb - 1) / (16 * mb);
                groups = ((int) sgrid.y + nb - 1) / nb;
                ndirs = (int) sgrid.z;
                ks = step_slices(P.N, groups, ndirs, nb, mb, cus);
#ifdef ASG_DEV_PROBES
                if (const char *ev = getenv("ASG_STEP_KS")) ks = atoi(ev) >= 1 && atoi(ev) <= kStepMaxSlices ? atoi(ev) : ks;
                if (getenv("ASG_STEP_SHOW")) fprintf(stderr, "[step grid] N=%d B=%d: %d row blocks, %d batch tile(s) per workgroup, %d slice(s), %d workgroups\n", P.N, P.B, mb, nb, ks, tiles * groups * ks * ndirs);
#endif
                sgrid = dim3((unsigned) ((tiles * ks * ndirs + 7) / 8 * 8 * groups), 1, 1);
                // matrix loads: non-temporal only when nobody else wants the tile (one batch tile group) and the directions' matrices do not
                // fit the memory-side cache.  T=200, B=32, nt / default, ms per step: N = 4000 8.46 / 7.57, 6000 (288 MB) 13.66 / 13.06,
                // 7000 (392 MB) 18.20 / 19.26, 8000 21.6 / 23.7, cfg 5 (800 MB) 133.9 / 140.1 us per frame; B = 64 (two groups sharing each
                // tile through the L2): N = 6000 25.8 / 23.0, 7000 39.4 / 36.4.
                nt = (groups == 1 && (double) ndirs * P.N * (double) W.npad * (step_bf3((int) e, P.B) ? 6.0 : 4.0) > 320e6) ? 1 : 0;
#ifdef ASG_DEV_PROBES
                if (const char *ev = getenv("ASG_STEP_NT")) nt = atoi(ev) ? 1 : 0;
#endif
            }
            if constexpr (StepUsesMfma<R>::v) {
                if (step_bf3((int) e, P.B)) {
                    for (int n = 0; n + 1 < P.T; ++n) {
#define ASG_BF3_LAUNCH(NB_, MB_) hipLaunchKernelGGL((fwd_step_bf3_kernel<NB_, MB_>), sgrid, dim3(256), 0, stream, P, Sd[0], Sd[1], n, do_a ? 0 : 1, ks, tiles, groups, ndirs, nt)
                        if (nb == 2) { if (mb == 2) ASG_BF3_LAUNCH(2, 2); else if (mb == 3) ASG_BF3_LAUNCH(2, 3); else if (mb == 4) ASG_BF3_LAUNCH(2, 4); else ASG_BF3_LAUNCH(2, 5); }
                        else { if (mb == 2) ASG_BF3_LAUNCH(1, 2); else if (mb == 3) ASG_BF3_LAUNCH(1, 3); else if (mb == 4) ASG_BF3_LAUNCH(1, 4); else ASG_BF3_LAUNCH(1, 5); }
#undef ASG_BF3_LAUNCH
                    }
                    stepped = true;
                }
            }
            for (int n = 0; !stepped && n + 1 < P.T; ++n) {
                if constexpr (StepUsesMfma<R>::v) {
#define ASG_STEP_LAUNCH(NB_, MB_, HALF_) hipLaunchKernelGGL((fwd_step_kernel<R, NB_, MB_, HALF_>), sgrid, dim3(256), 0, stream, P, Sd[0], Sd[1], n, do_a ? 0 : 1, ks, tiles, groups, ndirs, nt)
                    if (nb == 2) { if (mb == 2) ASG_STEP_LAUNCH(2, 2, false); else if (mb == 3) ASG_STEP_LAUNCH(2, 3, false); else if (mb == 4) ASG_STEP_LAUNCH(2, 4, false); else ASG_STEP_LAUNCH(2, 5, false); }
                    else if (half) { if (mb == 2) ASG_STEP_LAUNCH(1, 2, true); else if (mb == 3) ASG_STEP_LAUNCH(1, 3, true); else if (mb == 4) ASG_STEP_LAUNCH(1, 4, true); else ASG_STEP_LAUNCH(1, 5, true); }
                    else { if (mb == 2) ASG_STEP_LAUNCH(1, 2, false); else if (mb == 3) ASG_STEP_LAUNCH(1, 3, false); else if (mb == 4) ASG_STEP_LAUNCH(1, 4, false); else ASG_STEP_LAUNCH(1, 5, false); }
#undef ASG_STEP_LAUNCH
                } else {
                    hipLaunchKernelGGL((fwd_step_kernel<R, 1, kStepMB, false>), sgrid, dim3(256), 0, stream, P, Sd[0], Sd[1], n, do_a ? 0 : 1, ks, 0, 1, 1, 1);
                }
            }
        }
        if (do_b)
            hipLaunchKernelGGL((fwd_score_kernel<R, true>), dim3(P.B), dim3(256), 0, stream, P, Sd[1], (R *) O.full_scores);
        if (do_a && O.full_scores_alpha)
            hipLaunchKernelGGL((fwd_score_kernel<R, false>), dim3(P.B), dim3(256), 0, stream, P, Sd[0],
                               (R *) O.full_scores_alpha);
    }
    (void) store;
    return hipGetLastError();
}

template hipError_t launch_prep_generic<float>(const Problem &, const State &, hipStream_t);
template hipError_t launch_prep_generic<double>(const Problem &, const State &, hipStream_t);
template hipError_t launch_fwd_full_generic<float>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);
template hipError_t launch_fwd_full_generic<double>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);

}  // namespace asg
