// Host-side fast path of ASGLossFunction (torch_asg_amd/asg.py): the per-call work of HipBackend.loss_forward /
// loss_backward -- argument checks, the asg_problem block, three allocations, the call through the C ABI -- done in
// C++ instead of ~40 Python statements and two ctypes calls (tools/host_pieces2.py: 36 + 27 us -> see DESIGN.md 7).
//
// This file is plumbing above the C ABI, not part of it: it touches libasg_hip.so only through the function addresses
// that _lib.py resolved (so ASG_HIP_LIB variants are honoured) and decides nothing the Python path does not decide:
//   * anything unusual (CPU tensors, wrong dtypes or shapes, strided lengths, another current device, bf16 off the
//     fused route) makes try_loss_forward return None, and asg.py runs its own path, which converts or raises with the
//     messages the tests pin;
//   * the zeroed sync regions and the side-stream contexts stay owned by HipBackend (_sync / _context): they are asked
//     for through `host` once per (device, stream, capture) and remembered in the backend's own Fast object; HipBackend.release() calls reset().
// The reference's counterpart is its pybind layer, native/extension.cpp:15-29 + streamlined_fast_gpu.cpp:17-68.
#include <torch/extension.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>

#include <map>
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
};
constexpr int kSingleLaunch = ASG_FLAG_SINGLE_LAUNCH, kAlphaScores = ASG_FLAG_ALPHA_SCORES;

bool lengths_ok(const c10::optional<at::Tensor> &t, const at::Device &dev, int64_t B) {
    if (!t.has_value() || !t->defined()) return true;
    return t->scalar_type() == at::kLong && t->device() == dev && t->dim() == 1 && t->size(0) == B && t->is_contiguous();
}

// One per HipBackend (the backend owns it and is the only caller, so `host` is a borrowed reference: no cycle).
struct Fast {
Api api{};
py::handle host;                                                         // the HipBackend
std::map<std::tuple<int, void *, unsigned long long>, std::pair<void *, size_t>> sync_cache;
std::map<std::pair<int, void *>, void *> ctx_cache;
int cus_of[64] = {0};

void reset() {
    sync_cache.clear();
    ctx_cache.clear();
}

Fast(const std::vector<uint64_t> &a, py::handle backend) {
    TORCH_CHECK(a.size() == 11, "torch_asg_amd._binding.Fast: 11 addresses expected");
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
    host = backend;
}

void check(int status, const char *what) {
    TORCH_CHECK(status == 0, "torch_asg_amd: ", what, " failed: ", api.strerror_(status), " (status ", status, ")");
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

// HipBackend.fused_preferred: every XCD must hold three workgroups for each of its utterances
bool fused_preferred(int64_t B, int idx) {
    if (idx < 0 || idx >= 64) return false;
    if (!cus_of[idx]) cus_of[idx] = host.attr("_cu_count")(idx).cast<int>();
    const int64_t pairs = (B + 1) / 2;
    return ((pairs + 7) / 8) * 2 * 3 <= cus_of[idx] / 8;
}

void *sync_region(int idx, void *stream, size_t nbytes, const at::Device &dev) {
    unsigned long long cid = 0;
    check(api.capture_id(stream, &cid), "asg_stream_capture_id");
    const auto key = std::make_tuple(idx, stream, cid);
    auto it = sync_cache.find(key);
    if (it != sync_cache.end() && it->second.second >= nbytes) return it->second.first;
    if (sync_cache.size() > 4096) sync_cache.clear();
    py::object t = host.attr("_sync")(py::cast(dev), nbytes);                // zeroed, owned by the backend's pools
    const at::Tensor r = t.cast<at::Tensor>();
    sync_cache[key] = {r.data_ptr(), (size_t) r.numel()};
    return r.data_ptr();
}

void *context(int idx, void *stream, const at::Device &dev) {
    const auto key = std::make_pair(idx, stream);
    auto it = ctx_cache.find(key);
    if (it != ctx_cache.end()) return it->second;
    if (ctx_cache.size() > 32) ctx_cache.clear();
    py::object h = host.attr("_context")(py::cast(dev));                    // ctypes.c_void_p
    void *c = reinterpret_cast<void *>(h.attr("value").cast<uint64_t>());
    ctx_cache[key] = c;
    return c;
}

// -> None, or (loss, mode, buf0, buf1 | None, sc_bytes, state_bytes, scratch_bytes); mode 1 = fused step (buf0 = the
// workspace [scores | state | scratch], buf1 = grad_inputs), mode 0 = recursions only (buf0 = state)
py::object try_loss_forward(const at::Tensor &x, const at::Tensor &tr, const at::Tensor &tg,
                            const c10::optional<at::Tensor> &il, const c10::optional<at::Tensor> &tl, int red, int flags) {
    asg_problem p;
    if (!problem(p, x, tr, tg, il, tl)) return py::none();
    const at::Device dev = x.device();
    const int idx = dev.index();
    const bool use_fused = (flags & kSingleLaunch) && api.fused_supported(&p) && fused_preferred(p.B, idx);
    if (p.inputs_dtype && !use_fused) return py::none();
    const int64_t state_bytes = (int64_t) api.state_bytes(&p);
    void *stream = c10::hip::getCurrentHIPStream(idx).stream();
    const auto fopt = tr.options().requires_grad(false);
    at::Tensor loss = red == 0 ? at::empty({p.B}, fopt) : at::empty({}, fopt);
    const auto bopt = fopt.dtype(at::kByte);
    if (use_fused) {
        const int64_t fs = (int64_t) api.fused_scratch_bytes(&p);
        const size_t sync_bytes = api.fused_sync_bytes(&p);
        const int64_t sc_bytes = (2 * p.B * 4 + 255) / 256 * 256;
        at::Tensor ws = at::empty({sc_bytes + state_bytes + fs}, bopt);
        at::Tensor gin = at::empty({p.T, p.B, p.N}, x.options().requires_grad(false));
        char *base = (char *) ws.data_ptr();
        void *sync = sync_region(idx, stream, sync_bytes, dev);
        check(api.fused_forward(&p, base + sc_bytes, (size_t) state_bytes, red, loss.data_ptr(), base,
                                base + sc_bytes + state_bytes, (size_t) fs, gin.data_ptr(), sync, 0, stream),
              "asg_loss_fused_forward");
        return py::make_tuple(loss, 1, ws, gin, sc_bytes, state_bytes, fs);
    }
    at::Tensor state = at::empty({std::max<int64_t>(state_bytes, 256)}, bopt);
    at::Tensor scores = at::empty({2, p.B}, x.options().requires_grad(false));
    check(api.loss_forward((asg_ctx *) context(idx, stream, dev), &p, state.data_ptr(), (size_t) state.numel(), red,
                           loss.data_ptr(), scores.data_ptr(), flags & ~kAlphaScores, stream),
          "asg_loss_forward");
    return py::make_tuple(loss, 0, state, py::none(), 0, state_bytes, 0);
}

// -> None, or (grad_transition, grad_inputs)
// rec = (mode, sc_bytes, state_bytes, scratch_bytes, reduction) of the forward call
py::object try_loss_backward(const std::tuple<int, int64_t, int64_t, int64_t, int> &rec, const at::Tensor &buf0,
                             const c10::optional<at::Tensor> &buf1, const at::Tensor &grad_loss, const at::Tensor &x,
                             const at::Tensor &tr, const at::Tensor &tg, const c10::optional<at::Tensor> &il,
                             const c10::optional<at::Tensor> &tl) {
    const int mode = std::get<0>(rec), red = std::get<4>(rec);
    const int64_t sc_bytes = std::get<1>(rec), state_bytes = std::get<2>(rec), fs = std::get<3>(rec);
    asg_problem p;
    if (!problem(p, x, tr, tg, il, tl)) return py::none();
    const at::Device dev = x.device();
    if (grad_loss.device() != dev || grad_loss.scalar_type() != tr.scalar_type() || !grad_loss.is_contiguous() ||
        grad_loss.numel() != (red == 0 ? p.B : 1))
        return py::none();
    const int idx = dev.index();
    void *stream = c10::hip::getCurrentHIPStream(idx).stream();
    const auto fopt = tr.options().requires_grad(false);
    at::Tensor gtr = at::empty({p.N, p.N}, fopt);
    if (mode == 1) {
        if (!buf1.has_value() || !buf1->defined() || buf0.numel() < sc_bytes + state_bytes + fs) return py::none();
        char *base = (char *) buf0.data_ptr();
        check(api.fused_backward(&p, base + sc_bytes, (size_t) state_bytes, red, grad_loss.data_ptr(),
                                 base + sc_bytes + state_bytes, (size_t) fs, buf1->data_ptr(), gtr.data_ptr(), 0, stream),
              "asg_loss_fused_backward");
        return py::make_tuple(gtr, *buf1);
    }
    const int64_t scratch_bytes = std::max<int64_t>((int64_t) api.scratch_bytes(&p), 256);
    at::Tensor scratch = at::empty({scratch_bytes}, fopt.dtype(at::kByte));
    at::Tensor gin = at::empty({p.T, p.B, p.N}, x.options().requires_grad(false));
    check(api.loss_backward((asg_ctx *) context(idx, stream, dev), &p, buf0.data_ptr(), (size_t) buf0.numel(), red,
                            grad_loss.data_ptr(), scratch.data_ptr(), (size_t) scratch.numel(), gtr.data_ptr(),
                            gin.data_ptr(), 0, stream),
          "asg_loss_backward");
    return py::make_tuple(gtr, gin);
}

};  // struct Fast

}  // namespace

PYBIND11_MODULE(_binding, m) {
    m.doc() = "C++ fast path of torch_asg_amd.asg.ASGLossFunction above the C ABI of libasg_hip.so";
    py::class_<Fast>(m, "Fast")
        .def(py::init<const std::vector<uint64_t> &, py::handle>())
        .def("reset", &Fast::reset)
        .def("try_loss_forward", &Fast::try_loss_forward)
        .def("try_loss_backward", &Fast::try_loss_backward);
}
