// dexr_quad_inst.hip -- instantiation of the four-lanes-per-frame solve kernel (dexr_quad.hpp) for one joint bucket.
// Compile with -DDEXR_NMAX=<16|24>.
#include "dexr_quad.hpp"
#include "dexr_launch.hpp"

#ifndef DEXR_NMAX
#error "DEXR_NMAX not defined"
#endif

namespace dexr {
#define DEXR_QCAT_(a, b) a##b
#define DEXR_QCAT(a, b) DEXR_QCAT_(a, b)

hipError_t DEXR_QCAT(launch_quad_, DEXR_NMAX)(const KernelParams& kp, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static DynLds dyn;  // dynamic LDS above 64 KB: requested per kernel and per device (dexr_launch.hpp)
  hipError_t e = dyn.ensure(reinterpret_cast<const void*>(&dexr_quad_kernel<DEXR_NMAX>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dexr_quad_kernel<DEXR_NMAX>), grid, block, lds, st, kp, kp.comps);
  return hipGetLastError();
}
}  // namespace dexr
