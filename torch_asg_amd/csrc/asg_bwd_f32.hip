// torch_asg_amd/csrc/asg_bwd_f32.hip -- float instantiation of the stand-alone gradient-assembly kernels.
#define ASG_TU_R float
#include "asg_bwd_impl.inc"
