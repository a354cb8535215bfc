// kernels_fast_hyb.hip - the hybrid variant (candidate sets for shadow rays + hierarchy for extension rays) of kernels_fast.hip: the radiance kernels built with FMA contraction and f32 hardware
// approximations (PTGPU_MATH_FAST).  Compile with -ffp-contract=fast.
#include "device_params.h"
#define PT_STRICT 0
#define PT_BVH 2
#define PT_NS ptk_fast_hyb
#include "radiance_device.inc"
#include "radiance_kernels.inc"
