// dexr_red_inst.hip -- one instantiation of the reduced-variable solve kernel (dexr_red.hpp) per translation unit.
// Compile with -DDEXR_NV=<8|16>.
#include "dexr_launch.hpp"
#include "dexr_red.hpp"

#ifndef DEXR_NV
#error "DEXR_NV not defined"
#endif

namespace dexr {
#define DEXR_RCAT_(a, b) a##b
#define DEXR_RCAT(a, b) DEXR_RCAT_(a, b)
hipError_t DEXR_RCAT(launch_red_, DEXR_NV)(const KernelParams& kp, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static DynLds dyn;  // dynamic LDS above 64 KB: requested per kernel and per device (dexr_launch.hpp)
  hipError_t e = dyn.ensure(reinterpret_cast<const void*>(&dexr_red_kernel<DEXR_NV>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dexr_red_kernel<DEXR_NV>), grid, block, lds, st, kp, kp.comps);
  return hipGetLastError();
}
}  // namespace dexr
