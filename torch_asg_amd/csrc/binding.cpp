// Host-side fast path of the ASGLoss training step (torch_asg_amd/asg.py): the per-call work of
// HipBackend.loss_forward / loss_backward -- argument checks, the asg_problem block, three allocations, the call
// through the C ABI -- AND the autograd node, in C++ instead of a Python autograd.Function, ~40 Python statements and
// two ctypes calls (tools/host_pieces2.py, tools/autograd_floor.py: see DESIGN.md 6, "host cost").
//
// This file is plumbing above the C ABI, not part of it: it touches libasg_hip.so only through the function addresses
// that _lib.py resolved (so ASG_HIP_LIB variants are honoured) and decides nothing the Python path does not decide:
//   * anything unusual (CPU tensors, wrong dtypes or shapes, strided lengths, another current device, bf16 off the
//     fused route, a target axis longer than the time axis, a batch beyond the 32-bit offsets of the small path) makes
//     loss_apply / try_loss_forward return None, and asg.py runs its own path, which converts or raises with the
//     messages the tests pin;
//   * the zeroed sync regions and the side-stream contexts stay owned by HipBackend (_sync / _context): they are asked
//     for through `host` once per (device, stream, capture) and remembered in the backend's own Fast object;
//     HipBackend.release() calls reset().
// The reference's counterpart is its pybind layer, native/extension.cpp:15-29 + streamlined_fast_gpu.cpp:17-68, and
// its autograd.Function ASGGPUFast (asg.py:71-97).
//
// Two surfaces:
//   Fast.loss_apply(inputs, transition, targets, input_lengths, target_lengths, reduction, flags) -> loss | None
//       the whole step: forward + an AsgLossNode (torch::autograd::Node) attached to the loss.  The node keeps what
//       backward needs in SavedVariables (saved-tensor hooks see them), runs without the GIL on the engine's thread,
//       recomputes the fused step when a retained graph is walked a second time.
//   Fast.eval_apply(same arguments) -> loss | None
//       the evaluation route (beta recursions, no gradient) as one call: asg_loss_forward_only.
//   Fast.try_loss_forward / try_loss_backward
//       the same two calls for the Python ASGLossFunction (kept as the route for whatever loss_apply declines and
//       for ASG_NO_CPP_NODE=1): tuples in, tuples out.
#include <torch/extension.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/saved_variable.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "../../include/asg_hip.h"

namespace {

using fn_bytes = size_t (*)(const asg_problem *);
struct Api {
    fn_bytes state_bytes, scratch_bytes, fused_scratch_bytes, fused_sync_bytes;
    int (*fused_supported)(const asg_problem *);
    int (*capture_id)(void *, unsigned long long *);
    const char *(*strerror_)(int);
    int (*loss_forward)(asg_ctx *, const asg_problem *, void *, size_t, int, void *, void *, int, void *);
    int (*loss_backward)(asg_ctx *, const asg_problem *, const void *, size_t, int, const void *, void *, size_t, void *,
                         void *, int, void *);
    int (*fused_forward)(const asg_problem *, void *, size_t, int, void *, void *, void *, size_t, void *, void *, int,
                         void *);
    int (*fused_backward)(const asg_problem *, void *, size_t, int, const void *, void *, size_t, void *, void *, int,
                          void *);
    unsigned (*cluster_timeouts)(void);
    int (*loss_forward_only)(asg_ctx *, const asg_problem *, void *, size_t, int, void *, void *, size_t, int, void *);
};
constexpr int kSingleLaunch = ASG_FLAG_SINGLE_LAUNCH, kAlphaScores = ASG_FLAG_ALPHA_SCORES;

bool lengths_ok(const c10::optional<at::Tensor> &t, const at::Device &dev, int64_t B) {
    if (!t.has_value() || !t->defined()) return true;
    return t->scalar_type() == at::kLong && t->device() == dev && t->dim() == 1 && t->size(0) == B && t->is_contiguous();
}

// what one forward call produced, and what the matching backward call needs to know about it
struct Step {
    at::Tensor loss, buf0, buf1;       // fused step: buf0 = workspace [scores | state | scratch], buf1 = grad_inputs; else buf0 = state
    int mode = 0;                      // 1 = fused step, 0 = recursions only
    int64_t sc_bytes = 0, state_bytes = 0, fs = 0;
};

struct ShapeKey {
    int64_t T, B, N, S;
    int32_t dtype, inputs_dtype;
    bool operator<(const ShapeKey &o) const {
        return std::tie(T, B, N, S, dtype, inputs_dtype) < std::tie(o.T, o.B, o.N, o.S, o.dtype, o.inputs_dtype);
    }
};
struct ShapeInfo {
    int64_t state_bytes, scratch_bytes, fused_scratch, sc_bytes;
    size_t sync_bytes;
    bool fused_shape;                  // asg_loss_fused_supported for contiguous emissions of this shape
};

// One per HipBackend (the backend owns it and is the only caller, so `host` is a borrowed reference: no cycle).
struct Fast : std::enable_shared_from_this<Fast> {
Api api{};
py::handle host;                                                         // the HipBackend
std::mutex mu;                                                           // the caches below (forward holds the GIL, backward does not)
std::map<std::tuple<int, void *, unsigned long long>, std::pair<void *, size_t>> sync_cache;
std::map<std::pair<int, void *>, void *> ctx_cache;
std::map<ShapeKey, ShapeInfo> shape_cache;
int cus_of[64] = {0};
std::atomic<unsigned> faults_seen{0};

void reset() {
    std::lock_guard<std::mutex> g(mu);
    sync_cache.clear();
    ctx_cache.clear();
}

Fast(const std::vector<uint64_t> &a, py::handle backend) {
    TORCH_CHECK(a.size() == 13, "torch_asg_amd._binding.Fast: 13 addresses expected");
    size_t i = 0;
    auto next = [&]() { return reinterpret_cast<void *>(a[i++]); };
    api.state_bytes = (fn_bytes) next();
    api.scratch_bytes = (fn_bytes) next();
    api.fused_scratch_bytes = (fn_bytes) next();
    api.fused_sync_bytes = (fn_bytes) next();
    api.fused_supported = (decltype(api.fused_supported)) next();
    api.capture_id = (decltype(api.capture_id)) next();
    api.strerror_ = (decltype(api.strerror_)) next();
    api.loss_forward = (decltype(api.loss_forward)) next();
    api.loss_backward = (decltype(api.loss_backward)) next();
    api.fused_forward = (decltype(api.fused_forward)) next();
    api.fused_backward = (decltype(api.fused_backward)) next();
    api.cluster_timeouts = (decltype(api.cluster_timeouts)) next();
    api.loss_forward_only = (decltype(api.loss_forward_only)) next();
    host = backend;
    faults_seen = api.cluster_timeouts();
}

void check(int status, const char *what) {
    TORCH_CHECK(status == 0, "torch_asg_amd: ", what, " failed: ", api.strerror_(status), " (status ", status, ")");
}

// Resident-slice route (N > 256): a launch that timed out repaired itself in stream (asg_generic.hip: fwd_repair_kernel), but the route
// is gone for the rest of the process -- HipBackend.check_faults says so once (a RuntimeWarning).  The counter is a host-pinned word.
void check_faults(int64_t N) {
    if (N <= 256) return;
    const unsigned n = api.cluster_timeouts();
    if (n == faults_seen.load(std::memory_order_relaxed)) return;
    faults_seen = n;
    py::gil_scoped_acquire gil;
    try {
        host.attr("check_faults")();
    } catch (py::error_already_set &e) {      // (warnings turned into errors by the caller's filters: not on this thread)
        e.discard_as_unraisable("torch_asg_amd check_faults");
    }
}

// Fills `p`; false = not the plain case (the Python path takes over).
bool problem(asg_problem &p, const at::Tensor &x, const at::Tensor &tr, const at::Tensor &tg,
             const c10::optional<at::Tensor> &il, const c10::optional<at::Tensor> &tl) {
    if (!x.is_cuda() || x.dim() != 3 || tr.dim() != 2 || tg.dim() != 2) return false;
    const at::Device dev = x.device();
    if (dev.index() != c10::hip::current_device()) return false;
    const auto xt = x.scalar_type(), tt = tr.scalar_type();
    const bool bf16 = xt == at::kBFloat16 && tt == at::kFloat;
    if (!(bf16 || ((xt == at::kFloat || xt == at::kDouble) && tt == xt))) return false;
    const int64_t T = x.size(0), B = x.size(1), N = x.size(2);
    if (tr.device() != dev || tr.size(0) != N || tr.size(1) != N) return false;
    if (tg.scalar_type() != at::kLong || tg.device() != dev || tg.size(0) != B || tg.size(1) < 1) return false;
    if (!lengths_ok(il, dev, B) || !lengths_ok(tl, dev, B)) return false;
    p.inputs = x.data_ptr();
    for (int k = 0; k < 3; ++k) p.inputs_strides[k] = x.stride(k);
    p.transition = tr.data_ptr();
    p.transition_strides[0] = tr.stride(0);
    p.transition_strides[1] = tr.stride(1);
    p.targets = (const int64_t *) tg.data_ptr();
    p.targets_strides[0] = tg.stride(0);
    p.targets_strides[1] = tg.stride(1);
    p.input_lengths = il.has_value() && il->defined() ? (const int64_t *) il->data_ptr() : nullptr;
    p.target_lengths = tl.has_value() && tl->defined() ? (const int64_t *) tl->data_ptr() : nullptr;
    p.T = T, p.B = B, p.N = N, p.S = tg.size(1);
    p.dtype = xt == at::kDouble ? ASG_DTYPE_F64 : ASG_DTYPE_F32;
    p.inputs_dtype = bf16 ? ASG_DTYPE_BF16 : 0;
    return true;
}

// sizes of a shape, asked of the library once (asg_state_bytes & co. read T, B, N, S and the dtypes only)
ShapeInfo shape_info(const asg_problem &p) {
    const ShapeKey key{p.T, p.B, p.N, p.S, p.dtype, p.inputs_dtype};
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = shape_cache.find(key);
        if (it != shape_cache.end()) return it->second;
    }
    ShapeInfo s{};
    s.state_bytes = (int64_t) api.state_bytes(&p);
    s.scratch_bytes = std::max<int64_t>((int64_t) api.scratch_bytes(&p), 256);
    s.fused_scratch = (int64_t) api.fused_scratch_bytes(&p);
    s.sync_bytes = api.fused_sync_bytes(&p);
    s.sc_bytes = (2 * p.B * 4 + 255) / 256 * 256;
    asg_problem q = p;                  // the stride-dependent part of asg_loss_fused_supported is asked per call (forward_impl)
    q.inputs_strides[0] = p.B * p.N, q.inputs_strides[1] = p.N, q.inputs_strides[2] = 1;
    s.fused_shape = api.fused_supported(&q) != 0;
    std::lock_guard<std::mutex> g(mu);
    if (shape_cache.size() > 256) shape_cache.clear();
    shape_cache[key] = s;
    return s;
}

// HipBackend.fused_preferred: every XCD must hold three workgroups for each of its utterances
bool fused_preferred(int64_t B, int idx) {
    if (idx < 0 || idx >= 64) return false;
    if (!cus_of[idx]) {
        py::gil_scoped_acquire gil;
        cus_of[idx] = host.attr("_cu_count")(idx).cast<int>();
    }
    const int64_t pairs = (B + 1) / 2;
    return ((pairs + 7) / 8) * 2 * 3 <= cus_of[idx] / 8;
}

void *sync_region(int idx, void *stream, size_t nbytes, const at::Device &dev) {
    unsigned long long cid = 0;
    check(api.capture_id(stream, &cid), "asg_stream_capture_id");
    const auto key = std::make_tuple(idx, stream, cid);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = sync_cache.find(key);
        if (it != sync_cache.end() && it->second.second >= nbytes) return it->second.first;
    }
    void *ptr;
    size_t got;
    {
        py::gil_scoped_acquire gil;       // (never while holding `mu`: the forward path takes them in the order GIL, mu)
        py::object t = host.attr("_sync")(py::cast(dev), nbytes);            // zeroed, owned by the backend's pools
        const at::Tensor r = t.cast<at::Tensor>();
        ptr = r.data_ptr();
        got = (size_t) r.numel();
    }
    std::lock_guard<std::mutex> g(mu);
    if (sync_cache.size() > 4096) sync_cache.clear();
    sync_cache[key] = {ptr, got};
    return ptr;
}

void *context(int idx, void *stream, const at::Device &dev) {
    const auto key = std::make_pair(idx, stream);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ctx_cache.find(key);
        if (it != ctx_cache.end()) return it->second;
    }
    void *c;
    {
        // HipBackend._context keys on the thread's current stream; backward runs on the engine's thread, whose current
        // stream the engine has already set to the forward's
        py::gil_scoped_acquire gil;
        py::object h = host.attr("_context")(py::cast(dev));                // ctypes.c_void_p
        c = reinterpret_cast<void *>(h.attr("value").cast<uint64_t>());
    }
    std::lock_guard<std::mutex> g(mu);
    if (ctx_cache.size() > 32) ctx_cache.clear();
    ctx_cache[key] = c;
    return c;
}

// ---- the two calls ---------------------------------------------------------------------------------------------
// false = not the plain case
bool forward_impl(Step &r, const at::Tensor &x, const at::Tensor &tr, const at::Tensor &tg,
                  const c10::optional<at::Tensor> &il, const c10::optional<at::Tensor> &tl, int red, int flags) {
    asg_problem p;
    if (red < 0 || red > 2 || !problem(p, x, tr, tg, il, tl)) return false;
    const at::Device dev = x.device();
    const int idx = dev.index();
    const ShapeInfo si = shape_info(p);
    const bool use_fused = (flags & kSingleLaunch) && si.fused_shape && fused_preferred(p.B, idx) &&
                           (x.is_contiguous() || api.fused_supported(&p));
    if (p.inputs_dtype && !use_fused) return false;
    check_faults(p.N);
    void *stream = c10::hip::getCurrentHIPStream(idx).stream();
    const auto fopt = tr.options().requires_grad(false);
    r.loss = red == 0 ? at::empty({p.B}, fopt) : at::empty({}, fopt);
    const auto bopt = fopt.dtype(at::kByte);
    r.state_bytes = si.state_bytes;
    if (use_fused) {
        r.mode = 1, r.sc_bytes = si.sc_bytes, r.fs = si.fused_scratch;
        r.buf0 = at::empty({r.sc_bytes + r.state_bytes + r.fs}, bopt);
        r.buf1 = at::empty({p.T, p.B, p.N}, x.options().requires_grad(false));
        char *base = (char *) r.buf0.data_ptr();
        void *sync = sync_region(idx, stream, si.sync_bytes, dev);
        check(api.fused_forward(&p, base + r.sc_bytes, (size_t) r.state_bytes, red, r.loss.data_ptr(), base,
                                base + r.sc_bytes + r.state_bytes, (size_t) r.fs, r.buf1.data_ptr(), sync, 0, stream),
              "asg_loss_fused_forward");
        return true;
    }
    r.mode = 0, r.sc_bytes = 0, r.fs = 0;
    r.buf0 = at::empty({std::max<int64_t>(r.state_bytes, 256)}, bopt);
    r.buf1 = at::Tensor();
    at::Tensor scores = at::empty({2, p.B}, x.options().requires_grad(false));
    check(api.loss_forward((asg_ctx *) context(idx, stream, dev), &p, r.buf0.data_ptr(), (size_t) r.buf0.numel(), red,
                           r.loss.data_ptr(), scores.data_ptr(), flags & ~kAlphaScores, stream),
          "asg_loss_forward");
    return true;
}

// false = not the plain case; `grad_loss` must already have the dtype of `tr`, live on its device and be contiguous
bool backward_impl(at::Tensor &gtr, at::Tensor &gin, int mode, int64_t sc_bytes, int64_t state_bytes, int64_t fs, int red,
                   const at::Tensor &buf0, const at::Tensor &buf1, const at::Tensor &grad_loss, const at::Tensor &x,
                   const at::Tensor &tr, const at::Tensor &tg, const c10::optional<at::Tensor> &il,
                   const c10::optional<at::Tensor> &tl) {
    asg_problem p;
    if (!problem(p, x, tr, tg, il, tl)) return false;
    const at::Device dev = x.device();
    if (grad_loss.device() != dev || grad_loss.scalar_type() != tr.scalar_type() || !grad_loss.is_contiguous() ||
        grad_loss.numel() != (red == 0 ? p.B : 1))
        return false;
    if (buf0.device() != dev) return false;
    const int idx = dev.index();
    void *stream = c10::hip::getCurrentHIPStream(idx).stream();
    const auto fopt = tr.options().requires_grad(false);
    if (mode == 1) {
        if (!buf1.defined() || buf1.device() != dev || buf0.numel() < sc_bytes + state_bytes + fs) return false;
        gtr = at::empty({p.N, p.N}, fopt);
        char *base = (char *) buf0.data_ptr();
        check(api.fused_backward(&p, base + sc_bytes, (size_t) state_bytes, red, grad_loss.data_ptr(),
                                 base + sc_bytes + state_bytes, (size_t) fs, buf1.data_ptr(), gtr.data_ptr(), 0, stream),
              "asg_loss_fused_backward");
        gin = buf1;
        return true;
    }
    const ShapeInfo si = shape_info(p);
    gtr = at::empty({p.N, p.N}, fopt);
    at::Tensor scratch = at::empty({si.scratch_bytes}, fopt.dtype(at::kByte));
    gin = at::empty({p.T, p.B, p.N}, x.options().requires_grad(false));
    check(api.loss_backward((asg_ctx *) context(idx, stream, dev), &p, buf0.data_ptr(), (size_t) buf0.numel(), red,
                            grad_loss.data_ptr(), scratch.data_ptr(), (size_t) scratch.numel(), gtr.data_ptr(),
                            gin.data_ptr(), 0, stream),
          "asg_loss_backward");
    return true;
}

// -> None, or (loss, mode, buf0, buf1 | None, sc_bytes, state_bytes, scratch_bytes)
py::object try_loss_forward(const at::Tensor &x, const at::Tensor &tr, const at::Tensor &tg,
                            const c10::optional<at::Tensor> &il, const c10::optional<at::Tensor> &tl, int red, int flags) {
    Step r;
    if (!forward_impl(r, x, tr, tg, il, tl, red, flags)) return py::none();
    if (r.mode) return py::make_tuple(r.loss, 1, r.buf0, r.buf1, r.sc_bytes, r.state_bytes, r.fs);
    return py::make_tuple(r.loss, 0, r.buf0, py::none(), 0, r.state_bytes, 0);
}

// -> None, or (grad_transition, grad_inputs)
// rec = (mode, sc_bytes, state_bytes, scratch_bytes, reduction) of the forward call
py::object try_loss_backward(const std::tuple<int, int64_t, int64_t, int64_t, int> &rec, const at::Tensor &buf0,
                             const c10::optional<at::Tensor> &buf1, const at::Tensor &grad_loss, const at::Tensor &x,
                             const at::Tensor &tr, const at::Tensor &tg, const c10::optional<at::Tensor> &il,
                             const c10::optional<at::Tensor> &tl) {
    at::Tensor gtr, gin;
    if (!backward_impl(gtr, gin, std::get<0>(rec), std::get<1>(rec), std::get<2>(rec), std::get<3>(rec), std::get<4>(rec), buf0,
                       buf1.has_value() ? *buf1 : at::Tensor(), grad_loss, x, tr, tg, il, tl))
        return py::none();
    return py::make_tuple(gtr, gin);
}

py::object loss_apply(const at::Tensor &x, const at::Tensor &tr, const at::Tensor &tg, const c10::optional<at::Tensor> &il,
                      const c10::optional<at::Tensor> &tl, int red, int flags);

// what ASGLoss.forward settles in Python before any native call and these entry points do not: missing lengths (defaults: an allocation +
// fill), S > T (truncation), batches beyond the 32-bit offsets of the small path (split)
static bool plain_call(const at::Tensor &x, const at::Tensor &tg, const c10::optional<at::Tensor> &il, const c10::optional<at::Tensor> &tl) {
    if (!il.has_value() || !tl.has_value() || !il->defined() || !tl->defined()) return false;
    if (x.dim() != 3 || tg.dim() != 2 || tg.size(1) > x.size(0)) return false;
    if (x.size(2) <= 64) {
        const double w = x.scalar_type() == at::kDouble ? 8.0 : 4.0;
        if ((double) x.size(0) * (double) std::max(x.size(2), tg.size(1)) * w * (double) x.size(1) >= 4294967296.0) return false;
    }
    return true;
}

// The evaluation route (module.eval() or forward_only=True; asg.py:129-131 -> ASGGPUFastForwardOnly, asg.py:58-68): beta recursions
// only, nothing saved, no autograd graph -- one call, the `full - aligned` and the reduction inside the kernels (asg_loss_forward_only).
// -> loss | None
py::object eval_apply(const at::Tensor &x, const at::Tensor &tr, const at::Tensor &tg, const c10::optional<at::Tensor> &il,
                      const c10::optional<at::Tensor> &tl, int red, int flags) {
    if (!plain_call(x, tg, il, tl)) return py::none();
    asg_problem p;
    if (red < 0 || red > 2 || !problem(p, x, tr, tg, il, tl) || p.inputs_dtype) return py::none();
    at::AutoGradMode no_grad(false);
    const at::Device dev = x.device();
    const int idx = dev.index();
    check_faults(p.N);
    void *stream = c10::hip::getCurrentHIPStream(idx).stream();
    const auto fopt = tr.options().requires_grad(false);
    const auto bopt = fopt.dtype(at::kByte);
    at::Tensor loss = red == 0 ? at::empty({p.B}, fopt) : at::empty({}, fopt);
    const int64_t e = p.dtype == ASG_DTYPE_F64 ? 8 : 4;
    const int64_t sbytes = (2 * p.B * e + 255) / 256 * 256 + 256;
    at::Tensor scores = at::empty({sbytes}, bopt), state;
    if (p.N > 64 || p.S > 64) state = at::empty({std::max<int64_t>(shape_info(p).state_bytes, 256)}, bopt);
    check(api.loss_forward_only((asg_ctx *) context(idx, stream, dev), &p, state.defined() ? state.data_ptr() : nullptr,
                                state.defined() ? (size_t) state.numel() : 0, red, loss.data_ptr(), scores.data_ptr(), (size_t) sbytes,
                                flags & ~kAlphaScores, stream),
          "asg_loss_forward_only");
    return py::cast(loss);
}

};  // struct Fast

// ---- the autograd node --------------------------------------------------------------------------------------------
// grad_fn of the loss: input of the node = grad_loss, outputs = (grad_inputs, grad_transition) -- the edge order of
// loss_apply's collect_next_edges(inputs, transition).
struct AsgLossNode : public torch::autograd::Node {
    std::shared_ptr<Fast> fast;
    torch::autograd::SavedVariable x, tr, tg, il, tl, buf0, buf1;
    int mode = 0, red = 0, flags = 0;
    int64_t sc_bytes = 0, state_bytes = 0, fs = 0;
    bool consumed = false;           // fused step: backward has run once; its buffers were handed to autograd and rescaled in place
    std::mutex node_mu;

    std::string name() const override { return "AsgLossBackward"; }

    void release_variables() override {
        std::lock_guard<std::mutex> g(node_mu);
        x.reset_data(); tr.reset_data(); tg.reset_data(); il.reset_data(); tl.reset_data(); buf0.reset_data(); buf1.reset_data();
    }

    torch::autograd::variable_list apply(torch::autograd::variable_list &&grads) override {
        std::lock_guard<std::mutex> g(node_mu);
        TORCH_CHECK(grads.size() == 1, "AsgLossBackward: one incoming gradient expected");
        if (!grads[0].defined()) return {at::Tensor(), at::Tensor()};
        at::AutoGradMode no_grad(false);
        auto self = shared_from_this();
        const at::Tensor X = x.unpack(self), TR = tr.unpack(self), TG = tg.unpack(self);
        const c10::optional<at::Tensor> IL = il.unpack(self), TL = tl.unpack(self);
        at::Tensor B0, B1;
        if (consumed && mode == 1) {
            // a retained graph walked a second time: the gradient buffers of the first pass belong to autograd now
            Step r;
            TORCH_CHECK(fast->forward_impl(r, X, TR, TG, IL, TL, red, flags) && r.mode == 1,
                        "torch_asg_amd: the fused step could not be recomputed for a second backward pass");
            B0 = r.buf0, B1 = r.buf1;
        } else {
            B0 = buf0.unpack(self);
            if (mode == 1) B1 = buf1.unpack(self);
        }
        at::Tensor G = grads[0];
        if (G.scalar_type() != TR.scalar_type() || G.device() != TR.device()) G = G.to(TR.options().requires_grad(false));
        if (!G.is_contiguous()) G = G.contiguous();
        at::Tensor gtr, gin;
        if (!fast->backward_impl(gtr, gin, mode, sc_bytes, state_bytes, fs, red, B0, B1, G, X, TR, TG, IL, TL))
            gin = slow_backward(B0, B1, G, X, TR, TG, IL, TL, gtr);
        consumed = true;
        return {gin, gtr};
    }

    // what came back through saved-tensor hooks is not the plain case any more (another device, strided ...): the Python
    // statements convert it (HipBackend.loss_backward_tensors) -- rare, correctness only
    at::Tensor slow_backward(const at::Tensor &B0, const at::Tensor &B1, const at::Tensor &G, const at::Tensor &X, const at::Tensor &TR,
                             const at::Tensor &TG, const c10::optional<at::Tensor> &IL, const c10::optional<at::Tensor> &TL,
                             at::Tensor &gtr) {
        py::gil_scoped_acquire gil;
        py::object r = fast->host.attr("loss_backward_tensors")(mode, sc_bytes, state_bytes, fs, red, B0,
                                                                B1.defined() ? py::cast(B1) : py::none(), G, X, TR, TG,
                                                                IL.has_value() ? py::cast(*IL) : py::none(),
                                                                TL.has_value() ? py::cast(*TL) : py::none());
        auto t = r.cast<std::tuple<at::Tensor, at::Tensor>>();
        gtr = std::get<0>(t);
        return std::get<1>(t);
    }
};

py::object Fast::loss_apply(const at::Tensor &x, const at::Tensor &tr, const at::Tensor &tg, const c10::optional<at::Tensor> &il,
                            const c10::optional<at::Tensor> &tl, int red, int flags) {
    if (!plain_call(x, tg, il, tl)) return py::none();
    Step r;
    {
        at::AutoGradMode no_grad(false);
        if (!forward_impl(r, x, tr, tg, il, tl, red, flags)) return py::none();
    }
    if (at::GradMode::is_enabled() && (x.requires_grad() || tr.requires_grad())) {
        std::shared_ptr<AsgLossNode> node(new AsgLossNode(), torch::autograd::deleteNode);
        node->fast = shared_from_this();
        node->set_next_edges(torch::autograd::collect_next_edges(x, tr));
        node->x = torch::autograd::SavedVariable(x, false);
        node->tr = torch::autograd::SavedVariable(tr, false);
        node->tg = torch::autograd::SavedVariable(tg, false);
        node->il = torch::autograd::SavedVariable(*il, false);
        node->tl = torch::autograd::SavedVariable(*tl, false);
        node->buf0 = torch::autograd::SavedVariable(r.buf0, false);
        if (r.mode) node->buf1 = torch::autograd::SavedVariable(r.buf1, false);
        node->mode = r.mode, node->red = red, node->flags = flags;
        node->sc_bytes = r.sc_bytes, node->state_bytes = r.state_bytes, node->fs = r.fs;
        torch::autograd::set_history(r.loss, node);
    }
    return py::cast(r.loss);
}

}  // namespace

PYBIND11_MODULE(_binding, m) {
    m.doc() = "C++ fast path of the ASGLoss training step (forward, autograd node, backward) above the C ABI of libasg_hip.so";
    py::class_<Fast, std::shared_ptr<Fast>>(m, "Fast")
        .def(py::init<const std::vector<uint64_t> &, py::handle>())
        .def("reset", &Fast::reset)
        .def("loss_apply", &Fast::loss_apply)
        .def("eval_apply", &Fast::eval_apply)
        .def("try_loss_forward", &Fast::try_loss_forward)
        .def("try_loss_backward", &Fast::try_loss_backward);
}
