// dexr_launch.hpp -- launcher declarations: one symbol per (bucket, precision, mode) instantiation
// (each lives in its own translation unit, see dexr_inst.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dexr_kernel.hpp"

namespace dexr {
typedef hipError_t (*launch_fn)(const KernelParams& kp, dim3 grid, dim3 block, size_t lds, hipStream_t st);

#define DEXR_DECL(N)                                                                          \
  hipError_t launch_##N##_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f32 solve */ \
  hipError_t launch_##N##_1_0(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f64 solve */ \
  hipError_t launch_##N##_1_1(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f64 eval  */ \
  hipError_t launch_##N##_1_2(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f64 fk    */
DEXR_DECL(4)
DEXR_DECL(8)
DEXR_DECL(16)
DEXR_DECL(24)
DEXR_DECL(32)
#undef DEXR_DECL

static inline launch_fn find_launcher(int bucket, int f64, int mode) {
#define DEXR_CASE(N)                                                  \
  if (bucket == N) {                                                  \
    if (mode == MODE_SOLVE) return f64 ? launch_##N##_1_0 : launch_##N##_0_0; \
    if (mode == MODE_EVAL) return launch_##N##_1_1;                   \
    if (mode == MODE_FK) return launch_##N##_1_2;                     \
  }
  DEXR_CASE(4) DEXR_CASE(8) DEXR_CASE(16) DEXR_CASE(24) DEXR_CASE(32)
#undef DEXR_CASE
  return nullptr;
}
}  // namespace dexr
