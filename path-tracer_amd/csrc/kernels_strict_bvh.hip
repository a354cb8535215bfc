// kernels_strict_bvh.hip - the BVH variant (PTGPU_OPT_BVH) of kernels_strict.hip: the radiance kernels built without contraction and with f64 wherever the
// reference's host build promotes to double (PTGPU_MATH_STRICT).  Compile with -ffp-contract=off.
#include "device_params.h"
#define PT_STRICT 1
#define PT_BVH 1
#define PT_NS ptk_strict_bvh
#include "radiance_device.inc"
#include "radiance_kernels.inc"
