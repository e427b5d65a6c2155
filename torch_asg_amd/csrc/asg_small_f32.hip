// torch_asg_amd/csrc/asg_small_f32.hip -- float instantiation of the small-alphabet recursion kernels.
#define ASG_TU_R float
#include "asg_small_impl.inc"
