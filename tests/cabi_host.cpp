// tests/cabi_host.cpp -- a plain C++/HIP host (no Python, no torch) driving libasg_hip.so through include/asg_hip.h,
// the way a non-Python reference-side binding would, and checking the results against the
// CPU oracle's C entry points (test infrastructure: this file lives under tests/).
// Built and run by tests/test_cabi_host.py:
//   hipcc -O2 tests/cabi_host.cpp -Iinclude -Ltorch_asg_amd/csrc -lasg_hip -Loracle -lasg_oracle -o <tmp>/cabi_host
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "asg_hip.h"

extern "C" {
int asg_oracle_full_forward_f64(const double *, const int64_t *, const double *, const int64_t *, int64_t, int64_t, int64_t,
                                double *, double *, double *);
int asg_oracle_full_backward_f64(const double *, const double *, const double *, const double *, const int64_t *,
                                 const double *, int64_t, int64_t, int64_t, double *, double *);
int asg_oracle_aligned_forward_f64(const double *, const int64_t *, const int64_t *, const double *, const int64_t *,
                                   const int64_t *, int64_t, int64_t, int64_t, int64_t, double *, double *, double *);
int asg_oracle_aligned_backward_f64(const double *, const double *, const double *, const int64_t *, const double *,
                                    const int64_t *, const int64_t *, int64_t, int64_t, int64_t, int64_t, double *, double *);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int) e_, __FILE__, __LINE__); return 2; } } while (0)
#define AK(x) do { int s_ = (x); if (s_ != ASG_OK) { fprintf(stderr, "asg status %d (%s) at %s:%d\n", s_, asg_hip_strerror(s_), __FILE__, __LINE__); return 3; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double) (rng_state >> 11) / 9007199254740992.0; }

int main() {
    const int64_t T = 60, B = 5, N = 12, S = 7;
    std::vector<float> x(T * B * N), tr(N * N);
    std::vector<int64_t> tg(B * S), il(B), tl(B);
    for (auto &v : x) v = (float) (4.0 * urand() - 2.0);
    for (auto &v : tr) v = (float) urand();
    for (auto &v : tg) v = (int64_t) (urand() * N) % N;
    for (int64_t b = 0; b < B; ++b) { il[b] = T / 2 + (int64_t) (urand() * (T / 2)); tl[b] = 1 + (int64_t) (urand() * (S - 1)); }

    float *dx, *dtr, *dloss, *dscores, *dg, *dgt, *dgi; int64_t *dtg, *dil, *dtl; void *dstate, *dscratch;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dtr, tr.size() * 4)); CK(hipMalloc(&dtg, tg.size() * 8));
    CK(hipMalloc(&dil, B * 8)); CK(hipMalloc(&dtl, B * 8)); CK(hipMalloc(&dloss, B * 4)); CK(hipMalloc(&dscores, 2 * B * 4));
    CK(hipMalloc(&dg, B * 4)); CK(hipMalloc(&dgt, N * N * 4)); CK(hipMalloc(&dgi, T * B * N * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtr, tr.data(), tr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtg, tg.data(), tg.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dil, il.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtl, tl.data(), B * 8, hipMemcpyHostToDevice));
    std::vector<float> ones(B, 1.0f);
    CK(hipMemcpy(dg, ones.data(), B * 4, hipMemcpyHostToDevice));

    asg_problem p{};
    p.inputs = dx; p.inputs_strides[0] = B * N; p.inputs_strides[1] = N; p.inputs_strides[2] = 1;
    p.transition = dtr; p.transition_strides[0] = N; p.transition_strides[1] = 1;
    p.targets = dtg; p.targets_strides[0] = S; p.targets_strides[1] = 1;
    p.input_lengths = dil; p.target_lengths = dtl;
    p.T = T; p.B = B; p.N = N; p.S = S; p.dtype = ASG_DTYPE_F32;

    if (asg_hip_version() != ASG_HIP_VERSION) { fprintf(stderr, "version mismatch\n"); return 4; }
    asg_ctx *ctx = nullptr;
    AK(asg_ctx_create(&ctx));
    const size_t sb = asg_state_bytes(&p), cb = asg_scratch_bytes(&p);
    CK(hipMalloc(&dstate, sb)); CK(hipMalloc(&dscratch, cb));
    hipStream_t st; CK(hipStreamCreate(&st));
    AK(asg_loss_forward(ctx, &p, dstate, sb, ASG_REDUCTION_NONE, dloss, dscores, ASG_FLAG_SINGLE_LAUNCH, st));
    AK(asg_loss_backward(ctx, &p, dstate, sb, ASG_REDUCTION_NONE, dg, dscratch, cb, dgt, dgi, 0, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> loss(B), gt(N * N), gi(T * B * N);
    CK(hipMemcpy(loss.data(), dloss, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gt.data(), dgt, N * N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gi.data(), dgi, gi.size() * 4, hipMemcpyDeviceToHost));
    // too-small workspace and null pointers are rejected, not dereferenced
    if (asg_loss_forward(ctx, &p, dstate, sb / 2, ASG_REDUCTION_NONE, dloss, dscores, 0, st) != ASG_ERR_WORKSPACE) return 5;
    if (asg_loss_forward(ctx, &p, nullptr, sb, ASG_REDUCTION_NONE, dloss, dscores, 0, st) != ASG_ERR_INVALID) return 6;

    // ---- oracle (fp64) on the same inputs
    std::vector<double> xd(x.begin(), x.end()), trd(tr.begin(), tr.end());
    const int64_t istr[3] = {B * N, N, 1};
    std::vector<double> fs(B), as(B), fa(T * B * N), fb(T * B * N), aa(T * B * S), ab(T * B * S);
    if (asg_oracle_full_forward_f64(xd.data(), istr, trd.data(), il.data(), T, B, N, fs.data(), fa.data(), fb.data())) return 7;
    if (asg_oracle_aligned_forward_f64(xd.data(), istr, tg.data(), trd.data(), il.data(), tl.data(), T, B, N, S, as.data(), aa.data(), ab.data())) return 7;
    std::vector<double> gp(B, 1.0), gm(B, -1.0), gt1(N * N), gi1(T * B * N), gt2(N * N), gi2(T * B * N);
    if (asg_oracle_full_backward_f64(gp.data(), fa.data(), fb.data(), xd.data(), istr, trd.data(), T, B, N, gt1.data(), gi1.data())) return 7;
    if (asg_oracle_aligned_backward_f64(gm.data(), aa.data(), ab.data(), tg.data(), trd.data(), il.data(), tl.data(), T, B, N, S, gt2.data(), gi2.data())) return 7;
    double el = 0, eg = 0, et = 0, ml = 1, mg = 1, mt = 1;
    for (int64_t b = 0; b < B; ++b) { double r = fs[b] - as[b]; el = fmax(el, fabs(loss[b] - r)); ml = fmax(ml, fabs(r)); }
    for (size_t k = 0; k < gi.size(); ++k) { double r = gi1[k] + gi2[k]; eg = fmax(eg, fabs(gi[k] - r)); mg = fmax(mg, fabs(r)); }
    for (size_t k = 0; k < gt.size(); ++k) { double r = gt1[k] + gt2[k]; et = fmax(et, fabs(gt[k] - r)); mt = fmax(mt, fabs(r)); }
    printf("cabi_host: scaled max errors vs oracle  loss %.3e  grad_inputs %.3e  grad_transition %.3e\n", el / ml, eg / mg, et / mt);
    AK(asg_ctx_destroy(ctx));
    return (el / ml <= 1e-4 && eg / mg <= 1e-4 && et / mt <= 1e-4) ? 0 : 1;
}
