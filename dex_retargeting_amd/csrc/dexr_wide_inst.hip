// dexr_wide_inst.hip -- instantiation of the sixteen-lanes-per-frame solve kernel (dexr_wide.hpp) for one joint bucket.
// Compile with -DDEXR_NMAX=<16|24|32>.
#include "dexr_wide.hpp"
#include "dexr_launch.hpp"

#ifndef DEXR_NMAX
#error "DEXR_NMAX not defined"
#endif

namespace dexr {
#define DEXR_WCAT_(a, b) a##b
#define DEXR_WCAT(a, b) DEXR_WCAT_(a, b)

hipError_t DEXR_WCAT(launch_wide_, DEXR_NMAX)(const KernelParams& kp, const WideTable* wt, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static size_t configured = 0;  // dynamic LDS above 64 KB has to be requested once per kernel
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dexr_wide_kernel<DEXR_NMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  hipLaunchKernelGGL((dexr_wide_kernel<DEXR_NMAX>), grid, block, lds, st, kp, kp.comps, wt);
  return hipGetLastError();
}
size_t DEXR_WCAT(wide_lds_per_wave_, DEXR_NMAX)() { return (size_t)WideLds<DEXR_NMAX>::WAVE; }
}  // namespace dexr
