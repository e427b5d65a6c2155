// torch_asg_amd/csrc/asg_kernels.h -- parameter blocks + launch prototypes shared by the
// kernel translation units and the C-ABI (asg_api.hip).  Internal; the public surface is
// include/asg_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asg {

// One ASG problem instance as the caller handed it over (device pointers, element strides).
struct Problem {
    const void *inputs;          // [T,B,N] emissions, any strides
    int64_t is0, is1, is2;
    const void *transition;      // [N,N], transition[i][j] = score of j -> i
    int64_t ts0, ts1;
    const int64_t *targets;      // [B,S]
    int64_t gs0, gs1;
    const int64_t *in_len;       // [B] or nullptr (= T)
    const int64_t *tg_len;       // [B] or nullptr (= S)
    int T, B, N, S;
    int in_bf16;                 // emissions are bfloat16 (fused training step only); strides stay in elements
};

// Saved lattice state (forward -> backward), all in log2 units and RELATIVE per frame.
//   ah  [B][T][N]  full-lattice alpha-hat          bh  [B][T][N]  full-lattice beta-hat
//   ab  [B][T][S]  aligned alpha-bar               bb  [B][T][S]  aligned beta-bar
//   ehat [N][npad] row-normalised exp2 of the transition matrix (written once per forward by the alpha pass of
//                  utterance 0; npad = N rounded up to 8 on the small path, to 4 on the generic path), rmax [N]
//   fhat/cmax      column-normalised twin (generic path only)
//   asu [B][S][2]  per target position: {Tr2[O_s][O_s], Tr2[O_s][O_{s-1}]}  (log-zero where undefined)
//   asi [B][S][2]  int32 {O_s, O_{s-1}}
// ScaleLog (small path, klog [B][T][2]): what the full-lattice alpha pass folded into frame t's emission factor besides the
// emission and the row maximum -- entry t = {zb, ex}: the block scale and the power-of-two exponent, such that
//     sum_j ehat[i][j] * 2^ah[t-1][j]  =  2^( ah[t][i] - (fma(I[t][i], log2 e, rmax[i] - zb) - ex) )       (t >= 1)
// holds for the stored states with the chain's own rounding of the bracket.  The gradient pass recovers the row sums of the
// forward recursion from it (bwd_mfma_kernel) instead of recomputing them.  zb = NaN: the frame was produced by the exact
// per-node code and has no such relation.  Entry 0 = {kScaleLogMark, 0}, written by every alpha pass that stores states.
constexpr float kScaleLogMark = 1.0f;

struct State {
    void *ah, *bh, *ab, *bb;
    void *klog;
    void *ehat, *fhat, *rmax, *cmax;
    void *etile, *ftile;   // generic path, fp32: ehat / fhat again in the MFMA step kernel's operand order (asg_generic.hip)
    void *asu;
    int *asi;
    void *dbg;
    unsigned *ticket;   // 256 B, zeroed per call: arrival counter of the in-kernel loss reduction
    void *work;      // generic path: forward work buffers (emission maxima, p vectors, normalisers, offsets)
    int npad;
};

struct FwdOut {
    void *full_scores;           // [B]  from the beta pass (as the reference: fully_connected_lattice.cpp:89)
    void *aligned_scores;        // [B]  from the beta pass (force_aligned_lattice.cpp:316)
    void *full_scores_alpha;     // [B] or nullptr: same score from the alpha pass (cross-check)
    void *aligned_scores_alpha;  // [B] or nullptr
    // optional in-kernel loss reduction (small path): the LAST of the `expected` beta passes to finish reduces
    // loss[b] = full[b] - aligned[b] (reduction: 0 none, 1 sum, 2 mean) -- no separate reduce launch
    void *loss;
    unsigned *counter;           // State::ticket of this call (zeroed on the stream before the launch)
    int reduction, expected;
    int no_store;                // 1: scores only (eval / forward-only route): no lattice state is written.  A run-time
                                 // flag (stores are dropped by a zero-sized buffer resource), not a second set of kernels
};

struct BwdArgs {
    const void *grad_full;       // d(loss)/d(full_scores):    element b read at index b*gstride, times gscale
    const void *grad_aligned;    // d(loss)/d(aligned_scores); nullptr + neg_aligned: = -grad_full (ASG loss)
    int gstride;                 // 1 = per-utterance gradients, 0 = one scalar broadcast (reduced loss)
    int neg_aligned;
    double gscale;               // 1, or 1/B for reduction='mean' 
    void *grad_inputs;           // [T,B,N] contiguous
    void *grad_transition;       // [N,N] contiguous
    void *scratch;               // partial tiles etc.
    int chunk;                   // frames per workgroup (small path)
    int nchunks;
    int unit_grad;               // 1: the upstream gradient is 1 for every utterance (grad_full is not read)
};

// Fused training step of the whole criterion (asg_fused.hip): the forward launch also assembles the gradients for an
// upstream gradient of 1; the backward launch scales them (nothing to do when the upstream gradient IS 1), reduces
// the per-utterance transition-gradient tiles in a fixed order and redoes flagged utterances exactly.
struct FusedArgs {
    void *loss;                  // [B] (reduction none) or [1]
    void *scores;                // [2][B]: full, aligned
    void *grad_inputs;           // [T,B,N] contiguous
    void *tiles;                 // [B][2][N][N] per-utterance transition-gradient tiles (times gscale): alpha-side, beta-side frames
    int *flags;                  // [B]: 1 = the fused path declined this utterance (range guard, very short, time-out)
    void *dump;                  // [2][B] scratch scores for the exact redo
    void *p2;                    // [B][T][S] aligned posteriors, aligned workgroup -> full workgroup
    void *edges;                 // [B][2][3][2][64] double: aligned edge posteriors (stay | arrive) of the alpha-/beta-side frames, per finisher
    void *ascore;                // [B] double: aligned scores (log2 units)
    void *fscore;                // [B] double: full-lattice scores (log2 units), beta workgroup -> closing workgroup
    void *xstate;                // [B][2][xstate_blocks(T)][2][64][4] first-half states each full chain hands to the other
    void *rows;                  // [T,B,N] fp32: where the forward launch leaves its rows -- grad_inputs itself, or (bfloat16
                                 // emissions: grad_inputs is bfloat16) a work buffer
    void *in32;                  // bfloat16 emissions only: [B][T][N] fp32 copies of the emissions of FLAGGED utterances, made by
                                 // the forward launch for the exact stand-alone code (which reads fp32)
    void *aoff;                  // [B][2][T/16 + 2][2] double: per-block offsets of the stored aligned states
    unsigned *sync;              // caller-zeroed, returned zeroed: 64 words (word 0 = arrival ticket of the loss reduction)
                                 // + 16 words per utterance (UttSync in asg_fused.hip)
    unsigned *ticket2;           // 256 B, zeroed by the forward launch for the backward launch
    const void *grad_loss;       // backward: [B] (none) or [1]
    void *grad_transition;       // backward: [N,N]
    int reduction;
    float gscale;                // 1/B for reduction mean, else 1
};

enum ChainBits { kFullAlpha = 1, kFullBeta = 2, kAlignedAlpha = 4, kAlignedBeta = 8 };

// ---- small path: N <= 64, S <= 64, one wavefront per chain -------------------------------
// Developer / test switches, read from the environment once (asg_api.hip; asg_reload_env() reads them again).  -1 = not set.
struct Knobs {
    int fork_in_capture, pair_min_b, bwd_rowsum, no_cluster, no_mid, no_tile_step, step_one_tile, step_row_blocks, step_full_tile, step_no_bf3, step_bf3_min_b;
    char aligned_kernel;          // first letter of ASG_ALIGNED_KERNEL, or 0
};
const Knobs &knobs();

// chain_mask selects which of the four recursions this launch runs.
template <typename R>
hipError_t launch_fwd_small(const Problem &P, const State &W, const FwdOut &O, int chain_mask, bool store,
                            hipStream_t stream);
template <typename R>
hipError_t launch_bwd_small(const Problem &P, const State &W, const BwdArgs &A, int parts, hipStream_t stream);
hipError_t launch_fused_forward(const Problem &P, const State &W, const FusedArgs &F, hipStream_t stream);
hipError_t launch_fused_backward(const Problem &P, const State &W, const FusedArgs &F, hipStream_t stream);
// loss[b] = full[b] - aligned[b], reduced: 0 = none ([B] out), 1 = sum, 2 = mean ([1] out); fixed-order tree
template <typename R>
hipError_t launch_loss_reduce(const void *full, const void *aligned, int B, int reduction, void *out, hipStream_t stream);

// ---- generic path: any N (p-vector in LDS), S <= 1024 ----------------------------------
template <typename R>
hipError_t launch_prep_generic(const Problem &P, const State &W, hipStream_t stream);
template <typename R>
hipError_t launch_fwd_generic(const Problem &P, const State &W, const FwdOut &O, int chain_mask, bool store,
                              hipStream_t stream);
// parts: 1 = full lattice (N > 64), 2 = aligned lattice, 4 = grad buffers already hold the full-lattice part
template <typename R>
hipError_t launch_bwd_generic(const Problem &P, const State &W, const BwdArgs &A, int parts, hipStream_t stream);

// ---- best-path force alignment (S <= 64): work = [B][T] 64-bit back-pointer masks
template <typename R>
hipError_t launch_viterbi_small(const Problem &P, void *work, void *scores, void *path, hipStream_t stream);

// launches of the resident-slice forward kernel (256 < N <= 2048) of this process whose bounded waits ran out (asg_generic.hip)
unsigned cluster_timeouts();

size_t bwd_scratch_bytes_small(int elem, int T, int B, int N, int S, int *chunk, int *nchunks);
size_t bwd_scratch_bytes_generic(int elem, int T, int B, int N, int S);
size_t fwd_work_bytes_generic(int elem, int T, int B, int N);
size_t step_tile_bytes_generic(int elem, int N, int B);      // one operand-order copy of the normalised transition matrix (0: not used)

}  // namespace asg
