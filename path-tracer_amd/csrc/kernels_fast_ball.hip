// kernels_fast_ball.hip - kernels_fast_hyb.hip with the point test of PTGPU_CULL_BALL compiled in
// (its own instance: the code costs the other scenes 3-4 % merely by being there).
// Compile with -ffp-contract=fast-honor-pragmas.
#include "device_params.h"
#define PT_STRICT 0
#define PT_BVH 2
#define PT_BALL 1
#define PT_NS ptk_fast_ball
#include "radiance_device.inc"
#include "radiance_kernels.inc"
