// kernels_strict_ball.hip - kernels_strict.hip with the point test of PTGPU_CULL_BALL compiled in:
// the instance that proves the mode against the oracle's replay, bit for bit.
// Compile with -ffp-contract=off.
#include "device_params.h"
#define PT_STRICT 1
#define PT_BVH 0
#define PT_BALL 1
#define PT_NS ptk_strict_ball
#include "radiance_device.inc"
#include "radiance_kernels.inc"
