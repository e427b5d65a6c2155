// Developer check (GPU): asg_outer.h against a host loop -- lane mapping of the swaps and of the 16x16x4 f32 MFMA.
//   hipcc --offload-arch=gfx950 -O2 -I torch_asg_amd/csrc tools/ubench/outer_check.hip -o /tmp/outer_check && /tmp/outer_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "asg_outer.h"
using namespace asg;
constexpr int NT = 4, F = 24;
__global__ void k(const float *u, const float *v, float *out) {     // u, v: [F][64]; out [64][64]
    const int lane = threadIdx.x;
    V4<float> acc[NT * NT];
    for (int q = 0; q < NT * NT; ++q) acc[q] = V4<float>{0, 0, 0, 0};
    for (int f = 0; f < F; f += 4) {
        float uu[4], vv[4];
        for (int kk = 0; kk < 4; ++kk) { uu[kk] = u[(f + kk) * 64 + lane]; vv[kk] = v[(f + kk) * 64 + lane]; }
        outer4_accumulate<NT>(uu, vv, acc);
    }
    for (int r = 0; r < NT; ++r) for (int c = 0; c < NT; ++c) for (int q = 0; q < 4; ++q)
        out[(16 * r + 4 * (lane >> 4) + q) * 64 + 16 * c + (lane & 15)] = acc[r * NT + c][q];
}
int main() {
    std::vector<float> u(F * 64), v(F * 64), o(64 * 64), ref(64 * 64, 0.f);
    for (auto &x : u) x = (float) rand() / RAND_MAX - 0.3f;
    for (auto &x : v) x = (float) rand() / RAND_MAX * 2.f - 0.7f;
    for (int f = 0; f < F; ++f) for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) ref[i * 64 + j] = fmaf(u[f * 64 + i], v[f * 64 + j], ref[i * 64 + j]);
    float *du, *dv, *dout;
    hipMalloc(&du, u.size() * 4); hipMalloc(&dv, v.size() * 4); hipMalloc(&dout, o.size() * 4);
    hipMemcpy(du, u.data(), u.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, du, dv, dout);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0; int nbit = 0;
    for (int q = 0; q < 64 * 64; ++q) { worst = fmax(worst, fabs(o[q] - ref[q])); nbit += (o[q] == ref[q]); }
    printf("outer_check: max abs diff %.3e, bit-identical %d / %d  (%s)\n", worst, nbit, 64 * 64, worst < 1e-5 ? "OK" : "MISMATCH");
    return worst < 1e-5 ? 0 : 1;
}
