// torch_asg_amd/csrc/asg_fused.hip -- ONE workgroup per utterance: all four recursions of the utterance co-resident on
// one compute unit (stage 1: chains only; the state still goes to HBM and the stand-alone assembly kernels follow).
#include "asg_assemble.h"

namespace asg {
namespace {

// Wave roles.  A workgroup's waves are dealt to the CU's four SIMDs cyclically, so waves with equal (index % 4) share a
// SIMD: the two recursion wavefronts (0, 1) share theirs only with their own light producers (4, 5); the heavy helpers
// sit on the other two SIMDs.
template <int NP, bool STORE>
__global__ void __launch_bounds__(768, 1) fwd_cohab_kernel(Problem P, State W, FwdOut O) {
    __shared__ DuoLds LA, LB;
    const int b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) {
        LA.e_prod = 0; LA.verdict = 0; LA.csum = 0; LA.main_done = 0; LA.kill = 0; LA.c_done = 0; LA.prod_done = 0;
        LB.e_prod = 0; LB.verdict = 0; LB.csum = 0; LB.main_done = 0; LB.kill = 0; LB.c_done = 0; LB.prod_done = 0;
    }
    for (int q = threadIdx.x; q < kRing * 64; q += 768) {
        (&LA.s[0][0])[q] = __uint_as_float(kSentinel);
        (&LB.s[0][0])[q] = __uint_as_float(kSentinel);
    }
    __syncthreads();
#ifdef ASG_PROBE
    if ((threadIdx.x & 63) == 0 && b == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        ((unsigned *) W.dbg)[64 + wave] = hw;
    }
#endif
    switch (wave) {
        case 0: __builtin_amdgcn_s_setprio(3); duo_main<NP, STORE, false>(P, W, O, b, LA); break;
        case 1: __builtin_amdgcn_s_setprio(3); duo_main<NP, STORE, true>(P, W, O, b, LB); break;
        case 4: duo_producer<NP, false>(P, b, LA); break;
        case 5: duo_producer<NP, true>(P, b, LB); break;
        case 2: duo_consumer<NP, STORE, false>(P, W, O, b, LA); break;
        case 3: duo_consumer<NP, STORE, true>(P, W, O, b, LB); break;
        case 6: aligned_alpha_chain<float, STORE>(P, W, O, b); break;
        case 7: aligned_beta_chain<float, STORE>(P, W, O, b); break;
        default: break;
    }
}

template <int NP>
hipError_t launch_cohab_np(const Problem &P, const State &W, const FwdOut &O, bool store, hipStream_t st) {
    dim3 grid(P.B), block(768);
    if (store) hipLaunchKernelGGL((fwd_cohab_kernel<NP, true>), grid, block, 0, st, P, W, O);
    else hipLaunchKernelGGL((fwd_cohab_kernel<NP, false>), grid, block, 0, st, P, W, O);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_fwd_cohab(const Problem &P, const State &W, const FwdOut &O, bool store, hipStream_t stream) {
    const int N = P.N;
#ifdef ASG_DEV_ONLY_NP
    (void) N;
    return launch_cohab_np<ASG_DEV_ONLY_NP>(P, W, O, store, stream);
#else
    if (N <= 8) return launch_cohab_np<8>(P, W, O, store, stream);
    if (N <= 16) return launch_cohab_np<16>(P, W, O, store, stream);
    if (N <= 24) return launch_cohab_np<24>(P, W, O, store, stream);
    if (N <= 32) return launch_cohab_np<32>(P, W, O, store, stream);
    if (N <= 40) return launch_cohab_np<40>(P, W, O, store, stream);
    if (N <= 48) return launch_cohab_np<48>(P, W, O, store, stream);
    return launch_cohab_np<56>(P, W, O, store, stream);
#endif
}

}  // namespace asg
