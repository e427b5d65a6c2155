// torch_asg_amd/csrc/asg_batched.hip -- large batches (fp32, N <= 64): the full-lattice recursions sixteen utterances per
// workgroup on the matrix cores (device code: asg_batched.h).  One kernel, one launcher; the per-utterance kernels of
// asg_small_f32.hip follow on the stream for the aligned lattice and for the utterances this kernel flags.
#include "asg_batched.h"

namespace asg {
namespace {

// grid = ceil(B / 16) groups x (1 or 2 directions), block = 64 * NT.  Beta first: it owns the scores.
template <int NP, int MODE>
__global__ void __launch_bounds__(64 * ((NP + 15) / 16), 2) fwd_x16_kernel(Problem P, State W, FwdOut O, int chain_mask, int ngrp) {
    __shared__ X16Lds L;
    const bool both = (chain_mask & kFullAlpha) && (chain_mask & kFullBeta);
    const int grp = (int) blockIdx.x % ngrp;
    const bool beta = both ? (int) blockIdx.x < ngrp : (chain_mask & kFullBeta) != 0;
    if (beta) full_chain_x16<NP, true, MODE>(P, W, O, grp, W.xflags, L);
    else full_chain_x16<NP, false, MODE>(P, W, O, grp, W.xflags, L);
}

template <int NP>
hipError_t launch_np(const Problem &P, const State &W, const FwdOut &O, int fullm, int vec, hipStream_t st) {
    const int ngrp = (P.B + 15) / 16;
    const dim3 grid(ngrp * __builtin_popcount(fullm)), block(64 * ((NP + 15) / 16));
    // mode 2: the L1 norm comes out of the product's padding rows (needs a whole k-step of padding labels)
    if (vec && P.N % 16 != 0) hipLaunchKernelGGL((fwd_x16_kernel<NP, 2>), grid, block, 0, st, P, W, O, fullm, ngrp);
    else if (vec) hipLaunchKernelGGL((fwd_x16_kernel<NP, 1>), grid, block, 0, st, P, W, O, fullm, ngrp);
    else hipLaunchKernelGGL((fwd_x16_kernel<NP, 0>), grid, block, 0, st, P, W, O, fullm, ngrp);
    return hipGetLastError();
}

}  // namespace

// Batches from this many utterances up take the batched full-lattice chains (measured cross-over on MI355X; the
// environment variable is a developer / test knob).
int batched_min_batch() {
    const char *e = getenv("ASG_BATCHED_MIN_B");
    const int x = e ? atoi(e) : 0;
    return x > 0 ? x : (1 << 30);      // (off by default until it wins: see DESIGN.md)
}

bool batched_forward_applies(const Problem &P, const State &W, int chain_mask) {
    if (!(chain_mask & (kFullAlpha | kFullBeta)) || !W.xflags || P.N > 64 || P.B < batched_min_batch()) return false;
    if (P.is0 < 0 || P.is1 < 0 || P.is2 < 0) return false;
    // emissions through ONE buffer resource over the whole tensor (32-bit offsets), states with the "invalid = 2^31" trick
    const double span = ((double) (P.T - 1) * (double) P.is0 + (double) (P.B - 1) * (double) P.is1 + 63.0 * (double) P.is2 + 4.0) * 4.0;
    return span < 4294967000.0 && (double) P.B * P.T * P.N * 4.0 < 2147483648.0;
}

hipError_t launch_fwd_batched(const Problem &P, const State &W, const FwdOut &O, int chain_mask, hipStream_t stream) {
    const int fullm = chain_mask & (kFullAlpha | kFullBeta);
    const int vec = (P.N % 4 == 0 && P.is2 == 1 && P.is1 % 4 == 0 && P.is0 % 4 == 0 && ((uintptr_t) P.inputs & 15) == 0 &&
                     ((uintptr_t) W.ah & 15) == 0 && ((uintptr_t) W.bh & 15) == 0) ? 1 : 0;
#ifdef ASG_DEV_ONLY_NP
    return launch_np<ASG_DEV_ONLY_NP>(P, W, O, fullm, vec, stream);      // developer builds: one alphabet tile
#else
    const int N = P.N;
    if (N <= 8) return launch_np<8>(P, W, O, fullm, vec, stream);
    if (N <= 16) return launch_np<16>(P, W, O, fullm, vec, stream);
    if (N <= 24) return launch_np<24>(P, W, O, fullm, vec, stream);
    if (N <= 32) return launch_np<32>(P, W, O, fullm, vec, stream);
    if (N <= 40) return launch_np<40>(P, W, O, fullm, vec, stream);
    if (N <= 48) return launch_np<48>(P, W, O, fullm, vec, stream);
    if (N <= 56) return launch_np<56>(P, W, O, fullm, vec, stream);
    return launch_np<64>(P, W, O, fullm, vec, stream);
#endif
}

}  // namespace asg
