// dexr_big_inst.hip -- instantiation of the large-component solve kernel (dexr_big.hpp) for one joint bucket.
// Compile with -DDEXR_NMAX=<16|24|32>.
#include "dexr_big.hpp"
#include "dexr_launch.hpp"

#ifndef DEXR_NMAX
#error "DEXR_NMAX not defined"
#endif

namespace dexr {
#define DEXR_BCAT_(a, b) a##b
#define DEXR_BCAT(a, b) DEXR_BCAT_(a, b)

hipError_t DEXR_BCAT(launch_big_, DEXR_NMAX)(const KernelParams& kp, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static DynLds dyn;  // dynamic LDS above 64 KB: requested per kernel and per device (dexr_launch.hpp)
  hipError_t e = dyn.ensure(reinterpret_cast<const void*>(&dexr_big_kernel<DEXR_NMAX>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dexr_big_kernel<DEXR_NMAX>), grid, block, lds, st, kp, kp.comps);
  return hipGetLastError();
}
}  // namespace dexr
