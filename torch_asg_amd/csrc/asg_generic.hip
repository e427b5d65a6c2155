// torch_asg_amd/csrc/asg_generic.hip -- generic (large alphabet / long target) path. STUB for now.
#include "asg_common.h"
#include "asg_kernels.h"

namespace asg {

template <typename R>
hipError_t launch_prep_generic(const Problem &, const State &, hipStream_t) { return hipErrorNotSupported; }
template <typename R>
hipError_t launch_fwd_generic(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t) { return hipErrorNotSupported; }
template <typename R>
hipError_t launch_bwd_generic(const Problem &, const State &, const BwdArgs &, int, hipStream_t) { return hipErrorNotSupported; }

size_t bwd_scratch_bytes_generic(int, int, int, int, int) { return 256; }

template hipError_t launch_prep_generic<float>(const Problem &, const State &, hipStream_t);
template hipError_t launch_prep_generic<double>(const Problem &, const State &, hipStream_t);
template hipError_t launch_fwd_generic<float>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);
template hipError_t launch_fwd_generic<double>(const Problem &, const State &, const FwdOut &, int, bool, hipStream_t);
template hipError_t launch_bwd_generic<float>(const Problem &, const State &, const BwdArgs &, int, hipStream_t);
template hipError_t launch_bwd_generic<double>(const Problem &, const State &, const BwdArgs &, int, hipStream_t);

}  // namespace asg
