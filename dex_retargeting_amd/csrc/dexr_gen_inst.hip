// dexr_gen_inst.hip -- instantiations of the general kernel (dexr_gen.hpp): solve / objective evaluation / forward
// kinematics for models described by the generic table format.
#include "dexr_gen.hpp"
#include "dexr_launch.hpp"

namespace dexr {
size_t gen_lds_bytes(const GenTab& tb) { return gen_lds_doubles(tb.nj, tb.nf, tb.nt, tb.nv, tb.nfam) * sizeof(double); }

template <int MODE> static hipError_t launch_gen_mode(const KernelParams& kp, const GenTab& tb, dim3 grid, size_t lds, hipStream_t st) {
  static DynLds dyn;
  hipError_t e = dyn.ensure(reinterpret_cast<const void*>(&dexr_gen_kernel<MODE>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dexr_gen_kernel<MODE>), grid, dim3(64), lds, st, kp, tb);
  return hipGetLastError();
}

hipError_t launch_gen(int mode, const KernelParams& kp, const GenTab& tb, dim3 grid, size_t lds, hipStream_t st) {
  if (mode == MODE_SOLVE) return launch_gen_mode<MODE_SOLVE>(kp, tb, grid, lds, st);
  if (mode == MODE_EVAL) return launch_gen_mode<MODE_EVAL>(kp, tb, grid, lds, st);
  return launch_gen_mode<MODE_FK>(kp, tb, grid, lds, st);
}
}  // namespace dexr
