// dexr_wide_inst.hip -- instantiation of the sixteen-lanes-per-frame solve kernel (dexr_wide.hpp) for one joint bucket.
// Compile with -DDEXR_NMAX=<16|24|32> [-DDEXR_MIMIC=1: grid of the optimised variables, mimic joints folded (NMAX 16);
// -DDEXR_MODCHOL=1 with it: modified Cholesky and its damping rules; -DDEXR_SPRINT=1: one frame per wave (small batches)].
#include "dexr_wide.hpp"
#include "dexr_launch.hpp"

#ifndef DEXR_NMAX
#error "DEXR_NMAX not defined"
#endif

#ifndef DEXR_MIMIC
#define DEXR_MIMIC 0
#endif

namespace dexr {
#ifndef DEXR_MODCHOL
#define DEXR_MODCHOL 0
#endif
#ifndef DEXR_SPRINT
#define DEXR_SPRINT 0
#endif
#if DEXR_MIMIC && DEXR_MODCHOL && DEXR_SPRINT
#define DEXR_WNAME(base) base##s_mc_16
#elif DEXR_MIMIC && DEXR_SPRINT
#define DEXR_WNAME(base) base##s_m_16
#elif DEXR_MIMIC && DEXR_MODCHOL
#define DEXR_WNAME(base) base##mc_16
#elif DEXR_MIMIC
#define DEXR_WNAME(base) base##m_16
#elif DEXR_SPRINT
#define DEXR_WNAME(base) DEXR_WCAT(base##s_, DEXR_NMAX)
#else
#define DEXR_WNAME(base) DEXR_WCAT(base, DEXR_NMAX)
#endif
#define DEXR_WCAT_(a, b) a##b
#define DEXR_WCAT(a, b) DEXR_WCAT_(a, b)

hipError_t DEXR_WNAME(launch_wide_)(const KernelParams& kp, const WideTable* wt, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static DynLds dyn;  // dynamic LDS above 64 KB: requested per kernel and per device (dexr_launch.hpp)
  hipError_t e = dyn.ensure(reinterpret_cast<const void*>(&dexr_wide_kernel<DEXR_NMAX, (DEXR_MIMIC != 0), (DEXR_MODCHOL != 0), (DEXR_SPRINT != 0)>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dexr_wide_kernel<DEXR_NMAX, (DEXR_MIMIC != 0), (DEXR_MODCHOL != 0), (DEXR_SPRINT != 0)>), grid, block, lds, st, kp, kp.comps, wt);
  return hipGetLastError();
}
size_t DEXR_WNAME(wide_lds_per_wave_)() { return (size_t)WideLds<DEXR_NMAX, (DEXR_MIMIC != 0)>::WAVE; }
}  // namespace dexr
