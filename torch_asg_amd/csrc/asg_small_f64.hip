// torch_asg_amd/csrc/asg_small_f64.hip -- double instantiation of the small-alphabet recursion kernels.
#define ASG_TU_R double
#include "asg_small_impl.inc"
