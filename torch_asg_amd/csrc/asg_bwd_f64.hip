// torch_asg_amd/csrc/asg_bwd_f64.hip -- double instantiation of the stand-alone gradient-assembly kernels.
#define ASG_TU_R double
#include "asg_bwd_impl.inc"
