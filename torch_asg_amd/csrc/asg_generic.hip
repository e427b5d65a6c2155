// torch_asg_amd/csrc/asg_generic.hip -- generic ASG path: large alphabets (N > 64) and/or long targets
// (S > 64).  Same maths and same saved state as the small path (asg_small.hip), different parallelisation:
//
//  * full lattice, forward: TIME-SYNCHRONOUS and BATCH-COOPERATIVE.  One kernel launch per frame and
//    direction computes, for ALL utterances at once, s[i][b] = sum_j E[i][j] * p[b][j] as an LDS-tiled
//    [N x N] x [N x B] product (every tile of the normalised transition matrix E is read once per frame and
//    reused by the whole batch), then the per-node epilogue (log2, emission, lagged normaliser, exp2 for
//    the next frame).  This is the reference's per-time-step ATen loop
//    (/root/reference/torch_asg/native/fully_connected_lattice.cpp:22-28,44-46) with its ~10 launches per
//    step collapsed into one and its [B,N,N] temporaries never materialised.
//  * full lattice, gradient: two tiled products -- row sums for every (b,t) at once, then the outer-product
//    accumulation  gradTr = E o (U^T P)  -- instead of the reference's softmax over path_contrib
//    (fully_connected_lattice.cpp:49-63; that tensor would be 25.6 TB at T=2000 B=32 N=10000).
//  * aligned lattice: one workgroup per (utterance, direction), target axis spread over up to 16 waves,
//    neighbour exchange through LDS; gradient kernel per (utterance, frame-chunk).
//
// fp32 or fp64.  The recursion step of large alphabets IS a dense product and runs on the matrix instruction of the problem's precision
// (exact fp32 / fp64 MFMA; DESIGN.md section 5c says why that departs from the letter of "no MFMA").

// Round 6: the kernels live in three translation units (asg_generic_step.hip, asg_generic_aligned.hip, asg_generic_grad.hip; shared
// helpers and the per-unit launchers' declarations: asg_generic_common.h); this file is the routing between them.
#include "asg_generic_common.h"

namespace asg {

// chain_mask: which of the four recursions this call runs.  The aligned kernels are launched first (they are the short ones: the
// caller may have put them on a side stream), then the full-lattice route the alphabet selects:
//   64 < N <= 256 (fp32, 32-bit emission offsets)   fwd_mid_kernel, one launch for all frames
//   256 < N <= 2048 (fp64: <= 1024)                 fwd_cluster_kernel, one launch, the matrix resident in registers (+ its repair)
//   fp64 up to 2048                                 fwd_step_tile_kernel, a launch per frame
//   everything else (cfg 5)                         fwd_step_kernel, a launch per frame, the matrix streamed from HBM
template <typename R>
hipError_t launch_fwd_generic(const Problem &P, const State &W, const FwdOut &O, int chain_mask, bool store, hipStream_t stream) {
    const int full_mask = chain_mask & (kFullAlpha | kFullBeta);
    const int ali_mask = chain_mask & (kAlignedAlpha | kAlignedBeta);
    if (ali_mask) {
        const hipError_t e = launch_fwd_aligned_generic<R>(P, W, O, ali_mask, store, stream);
        if (e != hipSuccess) return e;
    }
    if (full_mask) return launch_fwd_full_generic<R>(P, W, O, full_mask, store, stream);
    return hipGetLastError();
}

// parts: 1 = full lattice (N > 64), 2 = aligned lattice, 4 = grad buffers already hold the full-lattice part
template <typename R>
hipError_t launch_bwd_generic(const Problem &P, const State &W, const BwdArgs &A0, int parts, hipStream_t stream) {
    BwdArgs A = A0;
    generic_chunks(P.T, P.B, &A.chunk, &A.nchunks);
    const GenericBwdLayout Y = generic_bwd_layout(sizeof(R), P, A);
    const bool do_full = parts & 1, do_ali = parts & 2, have_full = (parts & 5) != 0;
    bool fx_cleared = false;
    if (do_full) {
        const hipError_t e = launch_bwd_full_generic<R>(P, W, A, Y, do_ali, &fx_cleared, stream);
        if (e != hipSuccess) return e;
    }
    if (do_ali) return launch_bwd_aligned_generic<R>(P, W, A, Y, have_full, fx_cleared, stream);
    return hipGetLastError();
}

template hipError_t launch_fwd_generic<float>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);
template hipError_t launch_fwd_generic<double>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);
template hipError_t launch_bwd_generic<float>(const Problem &, const State &, const BwdArgs &, int, hipStream_t);
template hipError_t launch_bwd_generic<double>(const Problem &, const State &, const BwdArgs &, int, hipStream_t);

}  // namespace asg
