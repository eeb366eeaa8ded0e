// dexr_launch.hpp -- launcher declarations: one symbol per (bucket, precision, mode) instantiation
// (each lives in its own translation unit, see dexr_inst.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "dexr_kernel.hpp"

namespace dexr {
// Dynamic LDS above 64 KB has to be requested per kernel AND per device (hipFuncSetAttribute acts on the current device's
// copy of the function).  One DynLds per kernel instantiation caches the largest size granted so far on each device, under
// a lock: model handles are created and launched from several host threads, and one process may drive several GPUs.
struct DynLds {
  static constexpr int MAX_DEV = 64;
  size_t configured[MAX_DEV] = {};
  std::mutex mu;
  hipError_t ensure(const void* kernel, size_t lds) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    const bool known = dev >= 0 && dev < MAX_DEV;
    if (known && lds <= configured[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && known) configured[dev] = lds;
    return e;
  }
};

typedef hipError_t (*launch_fn)(const KernelParams& kp, dim3 grid, dim3 block, size_t lds, hipStream_t st);

#define DEXR_DECL_F32(N) hipError_t launch_##N##_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f32 solve */
DEXR_DECL_F32(4)
DEXR_DECL_F32(8)
DEXR_DECL_F32(16)
DEXR_DECL_F32(24)
#undef DEXR_DECL_F32
#define DEXR_DECL(N)                                                                          \
  hipError_t launch_##N##_1_0(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f64 solve */ \
  hipError_t launch_##N##_1_1(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f64 eval  */ \
  hipError_t launch_##N##_1_2(const KernelParams&, dim3, dim3, size_t, hipStream_t); /* f64 fk    */
DEXR_DECL(4)
DEXR_DECL(8)
DEXR_DECL(16)
DEXR_DECL(24)
DEXR_DECL(32)
#undef DEXR_DECL
// serial-chain specialisation (see LaneSolver's CHAIN flag): float32 solve, 4-joint bucket
hipError_t launch_chain_4_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
// small-component solve kernels with the extended addressing of KernelParams (fleet buckets, frame sequences); the
// buckets above 8 joints always carry it
hipError_t launch_ext_4_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_ext_8_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_ext_4_1_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_ext_8_1_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_ext_chain_4_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
// tip specialisation of the serial-chain kernel (dexr_tip.hpp): one vector term from a base frame to a frame on the last joint
hipError_t launch_tip_4_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_ext_tip_4_0_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_tip_4_1_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);      // the same pass in float64
hipError_t launch_ext_tip_4_1_0(const KernelParams&, dim3, dim3, size_t, hipStream_t);

// large-component kernel (dexr_big.hpp): float64 kinematics + float32 Hessian in LDS
hipError_t launch_big_16(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_big_24(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_big_32(const KernelParams&, dim3, dim3, size_t, hipStream_t);
static inline launch_fn find_big_launcher(int bucket) {
  return bucket == 16 ? launch_big_16 : bucket == 24 ? launch_big_24 : bucket == 32 ? launch_big_32 : nullptr;
}

// four-lanes-per-frame kernel for dense 9..24-joint components (dexr_quad.hpp)
hipError_t launch_quad_16(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_quad_24(const KernelParams&, dim3, dim3, size_t, hipStream_t);
static inline launch_fn find_quad_launcher(int bucket) {
  return bucket == 16 ? launch_quad_16 : bucket == 24 ? launch_quad_24 : nullptr;
}

// sixteen-lanes-per-frame kernel for dense 9..32-joint components without mimic joints (dexr_wide.hpp)
typedef hipError_t (*wide_launch_fn)(const KernelParams& kp, const WideTable* wt, dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t launch_wide_16(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);
hipError_t launch_wide_24(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);
hipError_t launch_wide_32(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);
hipError_t launch_wide_m_16(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);  // mimic joints folded
hipError_t launch_wide_mc_16(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);  // + modified Cholesky
hipError_t launch_wide_s_16(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);  // one frame per wave (SPRINT)
hipError_t launch_wide_s_24(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);
hipError_t launch_wide_s_32(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);
hipError_t launch_wide_s_m_16(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);
hipError_t launch_wide_s_mc_16(const KernelParams&, const WideTable*, dim3, dim3, size_t, hipStream_t);
size_t wide_lds_per_wave_s_m_16();
size_t wide_lds_per_wave_s_mc_16();
size_t wide_lds_per_wave_s_16();
size_t wide_lds_per_wave_s_24();
size_t wide_lds_per_wave_s_32();
size_t wide_lds_per_wave_m_16();
size_t wide_lds_per_wave_mc_16();
size_t wide_lds_per_wave_16();
size_t wide_lds_per_wave_24();
size_t wide_lds_per_wave_32();
static inline wide_launch_fn find_wide_launcher(int bucket) {
  return bucket == 16 ? launch_wide_16 : bucket == 24 ? launch_wide_24 : bucket == 32 ? launch_wide_32 : nullptr;
}
static inline wide_launch_fn find_wide_sprint_launcher(int bucket) {
  return bucket == 16 ? launch_wide_s_16 : bucket == 24 ? launch_wide_s_24 : bucket == 32 ? launch_wide_s_32 : nullptr;
}
static inline size_t wide_lds_per_wave(int bucket) {
  return bucket == 16 ? wide_lds_per_wave_16() : bucket == 24 ? wide_lds_per_wave_24() : bucket == 32 ? wide_lds_per_wave_32() : 0;
}

// reduced-variable kernel (dexr_red.hpp): Hessian of the n_var <= NV optimised variables in registers, kinematics in LDS
hipError_t launch_red_8(const KernelParams&, dim3, dim3, size_t, hipStream_t);
hipError_t launch_red_16(const KernelParams&, dim3, dim3, size_t, hipStream_t);
static inline launch_fn find_red_launcher(int nv_bucket) {
  return nv_bucket == 8 ? launch_red_8 : nv_bucket == 16 ? launch_red_16 : nullptr;
}

// general kernel (dexr_gen.hpp): models described by the generic table format
struct GenTab;
hipError_t launch_gen(int mode, const KernelParams& kp, const GenTab& tb, dim3 grid, size_t lds, hipStream_t st);
size_t gen_lds_bytes(const GenTab& tb);

static inline launch_fn find_launcher(int bucket, int f64, int mode, bool chain = false, bool ext = false, bool tip = false) {
  if (tip && chain && bucket == 4 && mode == MODE_SOLVE)
    return f64 ? (ext ? launch_ext_tip_4_1_0 : launch_tip_4_1_0) : (ext ? launch_ext_tip_4_0_0 : launch_tip_4_0_0);
  if (ext && mode == MODE_SOLVE && bucket <= 8) {
    if (chain && bucket == 4 && !f64) return launch_ext_chain_4_0_0;
    if (bucket == 4) return f64 ? launch_ext_4_1_0 : launch_ext_4_0_0;
    return f64 ? launch_ext_8_1_0 : launch_ext_8_0_0;
  }
  if (chain && bucket == 4 && !f64 && mode == MODE_SOLVE) return launch_chain_4_0_0;
#define DEXR_CASE(N)                                                  \
  if (bucket == N) {                                                  \
    if (mode == MODE_SOLVE && f64) return launch_##N##_1_0;            \
    if (mode == MODE_EVAL) return launch_##N##_1_1;                   \
    if (mode == MODE_FK) return launch_##N##_1_2;                     \
  }
  if (mode == MODE_SOLVE && !f64) {
    // the 32-joint bucket has no float32 instantiation (528-entry Hessian per lane does not fit the register file
    // in either precision; one kernel is enough there): float32 requests run the float64 kernel
    switch (bucket) {
      case 4: return launch_4_0_0;
      case 8: return launch_8_0_0;
      case 16: return launch_16_0_0;
      case 24: return launch_24_0_0;
      case 32: return launch_32_1_0;
    }
  }
  DEXR_CASE(4) DEXR_CASE(8) DEXR_CASE(16) DEXR_CASE(24) DEXR_CASE(32)
#undef DEXR_CASE
  return nullptr;
}
}  // namespace dexr
