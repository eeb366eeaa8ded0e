// dexr_inst.hip -- one kernel instantiation per translation unit so that the 20 (bucket, precision, mode)
// variants compile in parallel.  Compile with -DDEXR_NMAX=<4|8|16|24|32> -DDEXR_F64=<0|1> -DDEXR_MODE=<0|1|2>.
#include "dexr_launch.hpp"

#ifndef DEXR_NMAX
#error "DEXR_NMAX not defined"
#endif
#ifndef DEXR_CHAIN
#define DEXR_CHAIN 0
#endif
#ifndef DEXR_TIP
#define DEXR_TIP 0  // serial chains whose single term ends on the last joint: the pass of dexr_tip.hpp (needs DEXR_CHAIN)
#endif
#ifndef DEXR_EXT
#define DEXR_EXT 0  // small-component buckets only: the variant with fleet / sequence addressing (see KernelParams)
#endif

namespace dexr {
#if DEXR_F64
typedef double inst_real;
#else
typedef float inst_real;
#endif

#define DEXR_CAT_(a, b, c, d) a##b##_##c##_##d
#define DEXR_CAT(a, b, c, d) DEXR_CAT_(a, b, c, d)

#if DEXR_TIP && DEXR_EXT
#define DEXR_PREFIX launch_ext_tip_
#elif DEXR_TIP
#define DEXR_PREFIX launch_tip_
#elif DEXR_CHAIN && DEXR_EXT
#define DEXR_PREFIX launch_ext_chain_
#elif DEXR_CHAIN
#define DEXR_PREFIX launch_chain_
#elif DEXR_EXT
#define DEXR_PREFIX launch_ext_
#else
#define DEXR_PREFIX launch_
#endif

hipError_t DEXR_CAT(DEXR_PREFIX, DEXR_NMAX, DEXR_F64, DEXR_MODE)(const KernelParams& kp, dim3 grid, dim3 block, size_t lds,
                                                              hipStream_t st) {
  hipLaunchKernelGGL((dexr_kernel<DEXR_NMAX, inst_real, DEXR_MODE, (DEXR_CHAIN != 0), (DEXR_EXT != 0) || (DEXR_NMAX > 8), (DEXR_TIP != 0)>), grid, block, lds, st,
                     kp, kp.comps);
  return hipGetLastError();
}
}  // namespace dexr
