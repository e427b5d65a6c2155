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
    bool ok = el / ml <= 1e-4 && eg / mg <= 1e-4 && et / mt <= 1e-4;

    // ---- the fused training step (what ASGLoss runs at cfg 2 / 3): zeroed `sync`, the SAME state / scratch / grad_inputs
    // buffers to forward and backward, an upstream gradient that is not 1, called twice on the same sync region
    {
        if (!asg_loss_fused_supported(&p)) { fprintf(stderr, "fused step should take this problem\n"); return 8; }
        const size_t fs_bytes = asg_loss_fused_scratch_bytes(&p), sy_bytes = asg_loss_fused_sync_bytes(&p);
        void *dfs, *dsync; float *dl1, *dg1, *dgi2, *dgt2;
        CK(hipMalloc(&dfs, fs_bytes)); CK(hipMalloc(&dsync, sy_bytes)); CK(hipMemset(dsync, 0, sy_bytes));
        CK(hipMalloc(&dl1, 4)); CK(hipMalloc(&dg1, 4)); CK(hipMalloc(&dgi2, T * B * N * 4)); CK(hipMalloc(&dgt2, N * N * 4));
        const float up = 0.5f;
        CK(hipMemcpy(dg1, &up, 4, hipMemcpyHostToDevice));
        double worst = 0;
        for (int rep = 0; rep < 2; ++rep) {
            AK(asg_loss_fused_forward(&p, dstate, sb, ASG_REDUCTION_SUM, dl1, dscores, dfs, fs_bytes, dgi2, dsync, 0, st));
            AK(asg_loss_fused_backward(&p, dstate, sb, ASG_REDUCTION_SUM, dg1, dfs, fs_bytes, dgi2, dgt2, 0, st));
            CK(hipStreamSynchronize(st));
            float l1; std::vector<float> g2(T * B * N), t2(N * N);
            std::vector<unsigned char> sy(sy_bytes);
            CK(hipMemcpy(&l1, dl1, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g2.data(), dgi2, g2.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(t2.data(), dgt2, t2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sy.data(), dsync, sy_bytes, hipMemcpyDeviceToHost));
            for (unsigned char c : sy) if (c) { fprintf(stderr, "sync words not left zero\n"); return 9; }
            double ref = 0; for (int64_t b = 0; b < B; ++b) ref += fs[b] - as[b];
            double e1 = fabs(l1 - ref) / fmax(1.0, fabs(ref)), e2 = 0, e3 = 0;
            for (size_t k = 0; k < g2.size(); ++k) e2 = fmax(e2, fabs(g2[k] - up * (gi1[k] + gi2[k])));
            for (size_t k = 0; k < t2.size(); ++k) e3 = fmax(e3, fabs(t2[k] - up * (gt1[k] + gt2[k])));
            worst = fmax(worst, fmax(e1, fmax(e2 / mg, e3 / mt)));
        }
        printf("cabi_host: fused pair (reduction sum, upstream gradient 0.5, two calls on one sync region)  %.3e\n", worst);
        ok = ok && worst <= 1e-4;
        if (asg_loss_fused_forward(&p, dstate, sb, ASG_REDUCTION_SUM, dl1, dscores, dfs, fs_bytes / 2, dgi2, dsync, 0, st) != ASG_ERR_WORKSPACE) return 10;
        if (asg_loss_fused_forward(&p, dstate, sb, ASG_REDUCTION_SUM, dl1, dscores, dfs, fs_bytes, dgi2, nullptr, 0, st) != ASG_ERR_INVALID) return 10;
    }

    // ---- the entry points that map 1:1 onto the reference's pybind functions (extension.cpp:15-29):
    // fast_asg_gpu_forward / _backward / _forward_only and the serial pair of each lattice
    {
        float *dfull, *dali, *dgf, *dga;
        CK(hipMalloc(&dfull, 2 * B * 4)); CK(hipMalloc(&dali, 2 * B * 4)); CK(hipMalloc(&dgf, B * 4)); CK(hipMalloc(&dga, B * 4));
        std::vector<float> gf(B), ga(B);
        for (int64_t b = 0; b < B; ++b) { gf[b] = 1.0f + 0.25f * b; ga[b] = -0.5f - 0.125f * b; }
        CK(hipMemcpy(dgf, gf.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dga, ga.data(), B * 4, hipMemcpyHostToDevice));
        double worst = 0;
        for (int flags : {ASG_FLAG_STREAMS, ASG_FLAG_SINGLE_LAUNCH, 0}) {
            AK(asg_forward(ctx, &p, dstate, sb, dfull, dali, flags, st));
            AK(asg_backward(ctx, &p, dstate, sb, dgf, dga, dscratch, cb, dgt, dgi, 0, st));
            CK(hipStreamSynchronize(st));
            std::vector<float> f(B), a(B), g2(T * B * N), t2(N * N);
            CK(hipMemcpy(f.data(), dfull, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a.data(), dali, B * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(g2.data(), dgi, g2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(t2.data(), dgt, t2.size() * 4, hipMemcpyDeviceToHost));
            std::vector<double> gfd(gf.begin(), gf.end()), gad(ga.begin(), ga.end()), r1(N * N), r2(T * B * N), r3(N * N), r4(T * B * N);
            if (asg_oracle_full_backward_f64(gfd.data(), fa.data(), fb.data(), xd.data(), istr, trd.data(), T, B, N, r1.data(), r2.data())) return 7;
            if (asg_oracle_aligned_backward_f64(gad.data(), aa.data(), ab.data(), tg.data(), trd.data(), il.data(), tl.data(), T, B, N, S, r3.data(), r4.data())) return 7;
            double mf = 1, ma = 1, e = 0, mg2 = 1, mt2 = 1, eg2 = 0, et2 = 0;
            for (int64_t b = 0; b < B; ++b) { mf = fmax(mf, fabs(fs[b])); ma = fmax(ma, fabs(as[b])); }
            for (int64_t b = 0; b < B; ++b) e = fmax(e, fmax(fabs(f[b] - fs[b]) / mf, fabs(a[b] - as[b]) / ma));
            for (size_t k = 0; k < g2.size(); ++k) { double r = r2[k] + r4[k]; eg2 = fmax(eg2, fabs(g2[k] - r)); mg2 = fmax(mg2, fabs(r)); }
            for (size_t k = 0; k < t2.size(); ++k) { double r = r1[k] + r3[k]; et2 = fmax(et2, fabs(t2[k] - r)); mt2 = fmax(mt2, fabs(r)); }
            worst = fmax(worst, fmax(e, fmax(eg2 / mg2, et2 / mt2)));
            // evaluation route: scores only, no state
            AK(asg_forward_only(ctx, &p, nullptr, 0, dfull, dali, flags, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(f.data(), dfull, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a.data(), dali, B * 4, hipMemcpyDeviceToHost));
            for (int64_t b = 0; b < B; ++b) worst = fmax(worst, fmax(fabs(f[b] - fs[b]) / mf, fabs(a[b] - as[b]) / ma));
            // ... and the same route as ONE call with the loss reduced inside the kernels (asg_loss_forward_only), every reduction
            {
                asg_problem q = p;
                const size_t wb = asg_loss_forward_only_scores_bytes(&q);
                void *dwork; float *dl;
                CK(hipMalloc(&dwork, wb)); CK(hipMalloc(&dl, B * 4));
                for (int red = 0; red <= 2; ++red) {
                    AK(asg_loss_forward_only(ctx, &q, nullptr, 0, red, dl, dwork, wb, flags, st));
                    CK(hipStreamSynchronize(st));
                    std::vector<float> lv(B);
                    CK(hipMemcpy(lv.data(), dl, (red == 0 ? B : 1) * 4, hipMemcpyDeviceToHost));
                    double want = 0, scale = 1;
                    for (int64_t b = 0; b < B; ++b) { want += fs[b] - as[b]; scale = fmax(scale, fabs(fs[b] - as[b])); }
                    if (red == 0) { for (int64_t b = 0; b < B; ++b) worst = fmax(worst, fabs(lv[b] - (fs[b] - as[b])) / scale); }
                    else worst = fmax(worst, fabs(lv[0] - (red == 2 ? want / B : want)) / fmax(1.0, fabs(red == 2 ? want / B : want)));
                }
                if (asg_loss_forward_only(ctx, &q, nullptr, 0, 2, dl, dwork, wb - 1, flags, st) != ASG_ERR_WORKSPACE) { fprintf(stderr, "short work buffer accepted\n"); return 12; }
                CK(hipFree(dwork)); CK(hipFree(dl));
            }
        }
        // serial pair of each lattice (fully_connected_forward/backward, force_aligned_forward/backward)
        AK(asg_full_forward(&p, dstate, sb, dfull, 0, st));
        AK(asg_full_backward(&p, dstate, sb, dgf, dscratch, cb, dgt, dgi, st));
        CK(hipStreamSynchronize(st));
        {
            std::vector<float> g2(T * B * N), t2(N * N);
            CK(hipMemcpy(g2.data(), dgi, g2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(t2.data(), dgt, t2.size() * 4, hipMemcpyDeviceToHost));
            std::vector<double> gfd(gf.begin(), gf.end()), r1(N * N), r2(T * B * N);
            if (asg_oracle_full_backward_f64(gfd.data(), fa.data(), fb.data(), xd.data(), istr, trd.data(), T, B, N, r1.data(), r2.data())) return 7;
            double m1 = 1, m2 = 1, e1 = 0, e2 = 0;
            for (size_t k = 0; k < g2.size(); ++k) { e1 = fmax(e1, fabs(g2[k] - r2[k])); m1 = fmax(m1, fabs(r2[k])); }
            for (size_t k = 0; k < t2.size(); ++k) { e2 = fmax(e2, fabs(t2[k] - r1[k])); m2 = fmax(m2, fabs(r1[k])); }
            worst = fmax(worst, fmax(e1 / m1, e2 / m2));
        }
        AK(asg_aligned_forward(&p, dstate, sb, dali, 0, st));
        AK(asg_aligned_backward(&p, dstate, sb, dga, dscratch, cb, dgt, dgi, st));
        CK(hipStreamSynchronize(st));
        {
            std::vector<float> g2(T * B * N), t2(N * N);
            CK(hipMemcpy(g2.data(), dgi, g2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(t2.data(), dgt, t2.size() * 4, hipMemcpyDeviceToHost));
            std::vector<double> gad(ga.begin(), ga.end()), r3(N * N), r4(T * B * N);
            if (asg_oracle_aligned_backward_f64(gad.data(), aa.data(), ab.data(), tg.data(), trd.data(), il.data(), tl.data(), T, B, N, S, r3.data(), r4.data())) return 7;
            double m1 = 1, m2 = 1, e1 = 0, e2 = 0;
            for (size_t k = 0; k < g2.size(); ++k) { e1 = fmax(e1, fabs(g2[k] - r4[k])); m1 = fmax(m1, fabs(r4[k])); }
            for (size_t k = 0; k < t2.size(); ++k) { e2 = fmax(e2, fabs(t2[k] - r3[k])); m2 = fmax(m2, fabs(r3[k])); }
            worst = fmax(worst, fmax(e1 / m1, e2 / m2));
        }
        printf("cabi_host: asg_forward / asg_backward / asg_forward_only x 3 launch modes + the serial pairs  %.3e\n", worst);
        ok = ok && worst <= 1e-4;
        unsigned long long cid = 77;
        AK(asg_stream_capture_id(st, &cid));
        if (cid != 0) { fprintf(stderr, "stream is not capturing: id must be 0\n"); return 11; }
    }
    AK(asg_ctx_destroy(ctx));
    return ok ? 0 : 1;
}
