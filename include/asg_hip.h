/* include/asg_hip.h -- C ABI of libasg_hip.so: the MI355X (gfx950) ASG forward-backward hot path.
 *
 * Drop-in boundary for the native layer of zh217/torch-asg.  The reference binds its native code
 * through a pybind11 module `torch_asg_native` with seven functions
 * (/root/reference/torch_asg/native/extension.cpp:15-29); each entry point below names the one it
 * replaces.  Differences from the reference ABI, by design:
 *   - plain C: raw DEVICE pointers, explicit element strides, explicit sizes, explicit stream;
 *     no torch / ATen / pybind types anywhere;
 *   - the CALLER owns every buffer (state, scratch, outputs); the library never allocates device
 *     memory and never touches the thread's current stream;
 *   - no path_contrib tensors: the O(T*B*N*N) buffer of fully_connected_lattice.cpp:77 is never
 *     materialised -- the saved state is O(T*B*(N+S));
 *   - returns an int status (0 = ok) instead of throwing.
 *
 * All device pointers must be valid on the current HIP device.  Lengths/targets are int64
 * (the reference asserts kLong: utils.cpp:28,46).  `dtype` selects float32 / float64 for every
 * floating-point buffer of the call (utils.h:33-39 dispatch).
 */
#ifndef ASG_HIP_H
#define ASG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASG_HIP_VERSION 230

#define ASG_DTYPE_F32 0
#define ASG_DTYPE_F64 1
#define ASG_DTYPE_BF16 2            /* only as asg_problem::inputs_dtype (see there) */

/* status codes */
#define ASG_OK 0
#define ASG_ERR_INVALID 1       /* null pointer / bad dtype / negative size */
#define ASG_ERR_UNSUPPORTED 2   /* shape outside what this build supports */
#define ASG_ERR_WORKSPACE 3     /* state/scratch buffer too small */
#define ASG_ERR_HIP_BASE 1000   /* 1000 + hipError_t */

/* flags */
#define ASG_FLAG_STREAMS 1          /* run the full-lattice and force-aligned passes on two HIP streams
                                       (the reference's multi-stream route, streamlined_fast_gpu.cpp:104-230);
                                       without it everything is issued on `stream` in order
                                       (= ASGLoss(gpu_no_stream_impl=True), asg.py:124) */
#define ASG_FLAG_SINGLE_LAUNCH 2    /* all four recursions in ONE kernel launch (blockIdx.y = pass) */
#define ASG_FLAG_ALPHA_SCORES 8     /* debugging: forward also writes the scores obtained from the alpha passes
                                       into full_scores[B..2B) / aligned_scores[B..2B) */

/* One batch of utterances, as the reference's ASGLoss.forward receives it (asg.py:109). */
typedef struct asg_problem {
    const void *inputs;            /* [T,B,N] emissions (time-major), element strides below       */
    int64_t inputs_strides[3];
    const void *transition;        /* [N,N]; transition[i][j] = score of going from label j to i  */
    int64_t transition_strides[2];
    const int64_t *targets;        /* [B,S] int64                                                  */
    int64_t targets_strides[2];
    const int64_t *input_lengths;  /* [B] int64, or NULL = all T   (asg.py:116-117)               */
    const int64_t *target_lengths; /* [B] int64, or NULL = all S   (asg.py:113-114)               */
    int64_t T, B, N, S;
    int32_t dtype;                 /* ASG_DTYPE_*                                                  */
    int32_t inputs_dtype;          /* 0: emissions have `dtype`.  ASG_DTYPE_BF16 (with dtype = ASG_DTYPE_F32): emissions are
                                      bfloat16, everything else float32 (bf16 in, fp32 accumulate), and grad_inputs
                                      comes back as bfloat16.  Accepted by the asg_loss_fused_* pair only. */
} asg_problem;

/* Opaque context: the side stream + fork/join events of ASG_FLAG_STREAMS -- host handles only, no device memory and
 * no per-call state (the reference keeps none either: it borrows streams from the CUDA stream pool,
 * streamlined_fast_gpu.cpp:121-129).  Calls that use the SAME context are ordered through its events, so use one
 * context per calling stream (or thread); everything else in this library is re-entrant: whatever a call mutates on
 * the device lives in the buffers the caller passed to that call. */
typedef struct asg_ctx asg_ctx;

int asg_hip_version(void);
const char *asg_hip_strerror(int status);

/* Fault report (no reference counterpart): how many launches of the resident-slice forward kernel (256 < N <= 2048 in fp32, <= 1024 in
 * fp64: a grid of co-resident workgroups that wait for each other) of THIS process ran out of their bounded waits -- part of the grid
 * never became resident.  Such a call is REPAIRED IN STREAM: the launch raises a word in the call's own work area, and a repair kernel
 * that the same call enqueued behind it redoes the full-lattice recursion with no dependence between workgroups (exact; tens of
 * milliseconds; a no-op of ~2 us when nothing timed out) before the scores are computed -- the call's results are right, no NaN, no
 * error at a later call.  From then on the library takes the per-frame launches (no co-residency needed, 2-3x slower): this count is
 * what tells a caller that the fast route is gone.  Read from host-pinned memory, no synchronisation; the count of a call is visible once
 * that call has run. */
unsigned asg_cluster_timeouts(void);

/* Developer / test switches (ASG_FORK_IN_CAPTURE, ASG_PAIR_MIN_B, ASG_BWD_ROWSUM, ASG_NO_CLUSTER, ASG_NO_MID, ASG_NO_TILE_STEP, ASG_STEP_ONE_TILE, ASG_STEP_ROW_BLOCKS, ASG_STEP_FULL_TILE, ASG_STEP_NO_BF3, ASG_STEP_BF3_MIN_B,
 * ASG_ALIGNED_KERNEL) are read from the environment ONCE, at the first call that needs one -- no getenv on the per-call path.
 * A process that changes them afterwards (the test-suite does) calls this to have them read again.  No reference counterpart.
 * NOT for use while another thread is inside a call of this library: a forward call reads the switches more than once (the layout of
 * the operand-order matrices is chosen when they are built and again when they are read), and a reload between the two with another
 * ASG_STEP_* setting would make them disagree.  Each call keeps the previous block of switches alive (readers may still hold it): a few
 * dozen bytes per reload, never freed. */
void asg_reload_env(void);

int asg_ctx_create(asg_ctx **out);
int asg_ctx_destroy(asg_ctx *ctx);

/* *id = the identifier of the hipGraph capture `stream` is recording into, 0 when it is not capturing.  Host bindings
 * use it to give every capture its own `sync` region (asg_loss_fused_forward) without allocating under capture. */
int asg_stream_capture_id(void *stream, unsigned long long *id);

/* Bytes of saved lattice state (forward -> backward) and of backward scratch for a problem shape.
 * Only T,B,N,S,dtype of `p` are read.
 * The backward entry points take the SAME problem as their forward call: `inputs` and `transition` are read again (the
 * gradient pass recomputes the edge posteriors from the saved states and the emissions; nothing like the reference's
 * path_contrib is stored), so both tensors must still hold the values the forward call saw. */
size_t asg_state_bytes(const asg_problem *p);
size_t asg_scratch_bytes(const asg_problem *p);

/* ---- granular entry points: the reference's "serial" route (asg.py:124-128) ------------------- */

/* replaces torch_asg_native.fully_connected_forward (extension.cpp:16, fully_connected_lattice.cpp:65-91):
 * alpha+beta recursions of the fully-connected lattice; scores[B] = S_full. */
int asg_full_forward(const asg_problem *p, void *state, size_t state_bytes, void *scores, int flags, void *stream);

/* replaces torch_asg_native.fully_connected_backward (extension.cpp:17, fully_connected_lattice.cpp:93-105):
 * grad_transition[N,N], grad_inputs[T,B,N] (both contiguous, fully overwritten). */
int asg_full_backward(const asg_problem *p, const void *state, size_t state_bytes, const void *grad_out,
                      void *scratch, size_t scratch_bytes, void *grad_transition, void *grad_inputs, void *stream);

/* replaces torch_asg_native.force_aligned_forward (extension.cpp:18, force_aligned_lattice.cpp:266-319). */
int asg_aligned_forward(const asg_problem *p, void *state, size_t state_bytes, void *scores, int flags, void *stream);

/* replaces torch_asg_native.force_aligned_backward (extension.cpp:19, force_aligned_lattice.cpp:321-356). */
int asg_aligned_backward(const asg_problem *p, const void *state, size_t state_bytes, const void *grad_out,
                         void *scratch, size_t scratch_bytes, void *grad_transition, void *grad_inputs, void *stream);

/* ---- fused entry points: the reference's GPU fast route (asg.py:129-136) ---------------------- */

/* replaces torch_asg_native.fast_asg_gpu_forward (extension.cpp:25, streamlined_fast_gpu.cpp:104-230):
 * all four recursions; full_scores[B], aligned_scores[B] ([2B] each with ASG_FLAG_ALPHA_SCORES). */
int asg_forward(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes,
                void *full_scores, void *aligned_scores, int flags, void *stream);

/* replaces torch_asg_native.fast_asg_gpu_forward_only (extension.cpp:23, streamlined_fast_gpu.cpp:24-94):
 * beta recursions only, nothing saved.  `state` is only used as scratch by the large-alphabet path
 * (N > 64: the normalised transition matrices live there); it may be NULL when N <= 64 and S <= 64. */
int asg_forward_only(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes,
                     void *full_scores, void *aligned_scores, int flags, void *stream);

/* replaces torch_asg_native.fast_asg_gpu_backward (extension.cpp:27, streamlined_fast_gpu.cpp:236-297):
 * non-recursive gradient assembly for loss-side gradients grad_full[B], grad_aligned[B]. */
int asg_backward(asg_ctx *ctx, const asg_problem *p, const void *state, size_t state_bytes,
                 const void *grad_full, const void *grad_aligned, void *scratch, size_t scratch_bytes,
                 void *grad_transition, void *grad_inputs, int flags, void *stream);

/* ---- best-path (Viterbi) force alignment -- SURVEY.md 8(f)3.  No counterpart in the reference (README.md:33 lists
 * it as TODO; the lattice is force_aligned_lattice.cpp:84-111 with max instead of logsumexp,
 * doc/tech_report.tex:84-88).  scores[B] (dtype of inputs) = score of the best alignment, path[B][T] int64 = the
 * target POSITION occupied at each frame (-1 for frames >= input_lengths[b], and everywhere when the utterance has
 * no finite alignment: score -inf).  Tied comparisons keep "stay" (so among tied paths the one that advances
 * earliest is returned).  S <= 8192 like the other aligned-lattice entry points (beyond 1024 positions: N <= 38 400 in float32, 19 200 in float64).
 * `work` holds B*T*ceil(S/64) 64-bit back-pointer masks (asg_viterbi_work_bytes). */
size_t asg_viterbi_work_bytes(const asg_problem *p);
int asg_viterbi(asg_ctx *ctx, const asg_problem *p, void *work, size_t work_bytes, void *scores, int64_t *path,
                int flags, void *stream);

/* ---- whole-loss entry points (no counterpart in the reference's native layer: they fold the Python-side
 * `full - aligned` and reduction of asg.py:128,136-142 and their autograd into the kernels, so one ASGLoss
 * step is 2 + 2 kernel launches with no PyTorch glue kernels in between) ------------------------------------ */

#define ASG_REDUCTION_NONE 0
#define ASG_REDUCTION_SUM 1
#define ASG_REDUCTION_MEAN 2

/* loss = reduce_b(full[b] - aligned[b]); `loss` is [B] (none) or [1]; `scores` is a [2][B] work buffer that
 * receives full_scores then aligned_scores.
 * Alphabets of 257 .. 1024 labels (float32: .. 2048 while B <= 16) in every forward entry point: the full-lattice
 * recursions of all frames are ONE launch whose workgroups wait for each other frame by frame (the transition matrix
 * stays in their registers), sized to the device's compute units.  It therefore wants the device to itself: another
 * kernel that keeps compute units for seconds (a second process running the same route, say) can keep part of the
 * grid from starting.  A wait that runs out (~2^22 polls) ends the launch early and the repair kernel enqueued behind it by the
 * same call redoes the recursion without co-residency (asg_cluster_timeouts above): never a wrong number, never a NaN, never
 * a hang.  ASG_NO_CLUSTER=1 in the environment selects the launch-per-frame kernels from the start (2-3x slower,
 * no co-residency needed). */
int asg_loss_forward(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes, int reduction,
                     void *loss, void *scores, int flags, void *stream);

/* The evaluation route as ONE call: loss = reduce_b(full[b] - aligned[b]) from the beta recursions alone, nothing stored, no gradient
 * -- asg_forward_only (fast_asg_gpu_forward_only, streamlined_fast_gpu.cpp:24-94) with the `full - aligned` and the reduction of
 * asg.py:62-64,137-142 folded into the kernels, as asg_loss_forward does for the training route (small alphabets: the last beta
 * pass to finish reduces; large ones: one small reduction launch).  `scores`: asg_loss_forward_only_scores_bytes(p) bytes of work
 * space ([2][B] scores + 256 bytes for the arrival ticket).  `state` as for asg_forward_only (scratch of the large-alphabet path;
 * may be NULL when N <= 64 and S <= 64). */
size_t asg_loss_forward_only_scores_bytes(const asg_problem *p);
int asg_loss_forward_only(asg_ctx *ctx, const asg_problem *p, void *state, size_t state_bytes, int reduction,
                          void *loss, void *scores, size_t scores_bytes, int flags, void *stream);

/* gradients of the reduced loss: grad_loss is [B] (none) or [1]. */
int asg_loss_backward(asg_ctx *ctx, const asg_problem *p, const void *state, size_t state_bytes, int reduction,
                      const void *grad_loss, void *scratch, size_t scratch_bytes, void *grad_transition,
                      void *grad_inputs, int flags, void *stream);

/* ---- fused training step: the whole criterion, forward AND gradient assembly, in one launch -----------------
 * The reference's GPU fast route runs every recursion in forward and none in backward
 * (fast_asg_gpu_forward / fast_asg_gpu_backward, streamlined_fast_gpu.cpp:104-297); this pair goes one step further:
 * asg_loss_fused_forward also assembles, as the alpha and beta recursions cross, the full-lattice part of
 * d(loss)/d(inputs) and the per-utterance transition-gradient tiles (two per utterance: the frames each direction
 * reached second), and leaves the aligned posteriors beside them; asg_loss_fused_backward finishes the grad_inputs rows
 * (minus the aligned posteriors scattered to labels, times the actual upstream gradient), reduces the tiles in a fixed
 * order into grad_transition and redoes, exactly, any utterance the fused path declined (row sums outside the
 * fp32-safe range, fewer than 4 frames, a bounded wait on another workgroup that ran out).  Results are
 * bit-deterministic as long as no wait runs out, i.e. while the three workgroups of every utterance are co-resident
 * (an utterance redone by the exact code differs from the fused result in summation order, within the 1e-4 contract).
 * Only half of the lattice state ever goes to device memory.
 *   supported: float32, N < 64, S <= 64, T <= 4000 (asg_loss_fused_supported returns 1); otherwise use
 *              asg_loss_forward/backward.
 *   fast while: B <= 80 on 256 compute units -- the launch gives every utterance three compute units of its own and
 *              every XCD must hold three workgroups for each of its utterances; above that it still works (in rounds)
 *              but asg_loss_forward is faster, and the Python binding routes larger batches there.
 *   state:     asg_state_bytes(p) bytes, as for asg_loss_forward; the SAME buffer must be passed to backward.
 *   scratch:   asg_loss_fused_scratch_bytes(p) bytes; the SAME buffer must be passed to backward.
 *   grad_inputs [T,B,N] contiguous: partly written by forward, completed in place by backward.
 *   sync:      asg_loss_fused_sync_bytes(p) bytes of device memory that are ZERO on entry; the call leaves them
 *              zero.  They hold the words through which the workgroups of the launch talk to each other.  Calls that
 *              may run concurrently (different streams) need different regions; calls on one stream may share one.
 * The library allocates nothing and keeps no state between calls. */
int asg_loss_fused_supported(const asg_problem *p);
size_t asg_loss_fused_scratch_bytes(const asg_problem *p);
size_t asg_loss_fused_sync_bytes(const asg_problem *p);
int asg_loss_fused_forward(const asg_problem *p, void *state, size_t state_bytes, int reduction, void *loss, void *scores,
                           void *scratch, size_t scratch_bytes, void *grad_inputs, void *sync, int flags, void *stream);
int asg_loss_fused_backward(const asg_problem *p, void *state, size_t state_bytes, int reduction, const void *grad_loss,
                            void *scratch, size_t scratch_bytes, void *grad_inputs, void *grad_transition, int flags,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ASG_HIP_H */
