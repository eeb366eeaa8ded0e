// dexr_gen_inst.hip -- instantiations of the general kernel (dexr_gen.hpp): solve / objective evaluation / forward
// kinematics for models described by the generic table format.
#include "dexr_gen.hpp"
#include "dexr_launch.hpp"

namespace dexr {
size_t gen_lds_bytes(const GenTab& tb) { return gen_lds_doubles(tb.nj, tb.nf, tb.nt, tb.nv, tb.nfam) * sizeof(double); }

template <int MODE, int NI> static hipError_t launch_gen_mode(const KernelParams& kp, const GenTab& tb, dim3 grid, size_t lds, hipStream_t st) {
  static DynLds dyn;
  hipError_t e = dyn.ensure(reinterpret_cast<const void*>(&dexr_gen_kernel<MODE, NI>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dexr_gen_kernel<MODE, NI>), grid, dim3(64), lds, st, kp, tb);
  return hipGetLastError();
}

hipError_t launch_gen(int mode, const KernelParams& kp, const GenTab& tb, dim3 grid, size_t lds, hipStream_t st) {
  // the lower triangle of the Hessian is tiled over an 8 x 8 lane grid, NI x NI tiles per lane: NI = 5 serves models of up to
  // 40 variables at 2 waves per SIMD, the NI = 8 instantiation the rest (up to 64 variables)
  const bool small = tb.nv <= 8 * GEN_NI_SMALL;
  // (round 6) <= 24 variables: the NI = 3 instantiation of the solve -- 24 instead of 40 register rows in the factorisation, 6 instead
  // of 15 accumulator slots per lane
  if (mode == MODE_SOLVE && tb.nv <= 8 * GEN_NI_TINY) return launch_gen_mode<MODE_SOLVE, GEN_NI_TINY>(kp, tb, grid, lds, st);
  if (mode == MODE_SOLVE) return small ? launch_gen_mode<MODE_SOLVE, GEN_NI_SMALL>(kp, tb, grid, lds, st) : launch_gen_mode<MODE_SOLVE, GEN_NI_BIG>(kp, tb, grid, lds, st);
  if (mode == MODE_EVAL) return small ? launch_gen_mode<MODE_EVAL, GEN_NI_SMALL>(kp, tb, grid, lds, st) : launch_gen_mode<MODE_EVAL, GEN_NI_BIG>(kp, tb, grid, lds, st);
  return launch_gen_mode<MODE_FK, GEN_NI_SMALL>(kp, tb, grid, lds, st);
}
}  // namespace dexr
