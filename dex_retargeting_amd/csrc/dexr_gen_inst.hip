// dexr_gen_inst.hip -- instantiations of the general kernel (dexr_gen.hpp): solve / objective evaluation / forward
// kinematics for models described by the generic table format.
#include "dexr_gen.hpp"
#include "dexr_launch.hpp"

namespace dexr {
size_t gen_lds_bytes(const GenTab& tb) { return gen_lds_doubles(tb.nj, tb.nf, tb.nt, tb.nv, tb.nfam, tb.lt_in_lds != 0) * sizeof(double); }

template <int MODE, int NSLOT> static hipError_t launch_gen_mode(const KernelParams& kp, const GenTab& tb, dim3 grid, size_t lds, hipStream_t st) {
  static DynLds dyn;
  hipError_t e = dyn.ensure(reinterpret_cast<const void*>(&dexr_gen_kernel<MODE, NSLOT>), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dexr_gen_kernel<MODE, NSLOT>), grid, dim3(64), lds, st, kp, tb);
  return hipGetLastError();
}

hipError_t launch_gen(int mode, const KernelParams& kp, const GenTab& tb, dim3 grid, size_t lds, hipStream_t st) {
  // the lower triangle of the Hessian is accumulated in registers, NSLOT entries per lane: 12 x 64 entries serve models of
  // up to 38 variables at 2 waves per SIMD, the 33-slot instantiation the rest (up to 64 variables)
  const bool small = tb.nv * (tb.nv + 1) / 2 <= GEN_SLOTS_SMALL * 64;
  if (mode == MODE_SOLVE) return small ? launch_gen_mode<MODE_SOLVE, GEN_SLOTS_SMALL>(kp, tb, grid, lds, st) : launch_gen_mode<MODE_SOLVE, GEN_SLOTS_BIG>(kp, tb, grid, lds, st);
  if (mode == MODE_EVAL) return small ? launch_gen_mode<MODE_EVAL, GEN_SLOTS_SMALL>(kp, tb, grid, lds, st) : launch_gen_mode<MODE_EVAL, GEN_SLOTS_BIG>(kp, tb, grid, lds, st);
  return launch_gen_mode<MODE_FK, GEN_SLOTS_SMALL>(kp, tb, grid, lds, st);
}
}  // namespace dexr
