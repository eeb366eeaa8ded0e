// dexr_kernel.hpp -- hand-written HIP kernels for gfx950 (MI355X): batched retargeting solve.
//
// Mapping (MI355X-first):
//   * one LANE   = one (frame, component) sub-problem; all solver state (q, world axes/origins of the chain,
//     gradient, the dense lower-triangular Hessian and its Cholesky factor) lives in VGPRs with STATIC indices;
//   * one WAVE64 = 64 frames of the SAME component, so every table read (joint placements, masks, limits) is
//     wave-uniform -> scalar loads / SGPR operands, and every branch on the tree topology is a scalar branch;
//   * LDS holds only what needs a (uniform) dynamic index: frame positions and per-term targets/weights,
//     laid out [index][lane] so each ds_read/ds_write is conflict-free.
//
// What is computed follows the reference's objective closures
// (/root/reference/src/dex_retargeting/optimizer.py:138-200, 241-306, 456-577): forward kinematics of the
// chain, world-aligned point Jacobians J = a x (p - o), SmoothL1 of the vector norm (vector/dexpilot) or per
// coordinate (position), the mimic fold of kinematics_adaptor.py:102-113 and the 2*norm_delta*(x-last)
// regulariser.  The minimiser is ours: projected Levenberg-Marquardt/Newton with a register Cholesky.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dexr_tables.h"

#ifndef DEXR_TIP64_MINW
#define DEXR_TIP64_MINW 2  // float64 tip kernel: waves per SIMD requested (256 VGPRs)
#endif
#ifndef DEXR_BLOCK_MAX
#define DEXR_BLOCK_MAX 256  // threads per block of the register kernels (launch geometry: dexr_api.hip, DEXR_WPB waves)
#endif
#ifndef DEXR_TIP_BLOCK_MAX
#define DEXR_TIP_BLOCK_MAX 512  // the float32 tip kernel also runs in blocks of 8 waves (full-chip launches, see dexr_api.hip launch())
#endif
#ifndef DEXR_CHAIN_MINW
#define DEXR_CHAIN_MINW 4  // minimum waves per SIMD requested for the serial-chain kernel (caps its VGPR budget at 128)
#endif

namespace dexr {

struct KernelParams {
  const dexr_comp_table* __restrict__ comps;
  const float* __restrict__ ref;    // B x n_ref x 3
  const float* __restrict__ fixed;  // B x n_fixed
  const float* __restrict__ last;   // B x n_opt
  const float* x0;                  // B x n_opt start point (NULL: start from `last`); may alias qout
  const float* __restrict__ kpts;   // B x n_kp x 3 raw hand keypoints (NULL: `ref` holds ready-made ref_value rows)
  unsigned* queue;                  // n_comp work-queue heads (persistent-lane kernels), zeroed before the launch
  const double* __restrict__ xin;   // eval: B x n_opt ; fk: B x n_q
  uint32_t* state;                  // B
  float* qout;                      // B x n_opt
  double* qout64;                   // B x n_opt (optional)
  int32_t* status;                  // B
  int32_t* iters;                   // B
  float* fval;                      // B
  double* f64out;                   // eval: B ; fk: B x n_ref x 3
  double* g64out;                   // eval: B x n_opt
  int64_t B;
  int32_t n_comp, n_opt, n_fixed, n_ref, n_q, kind, num_fingers;
  float huber_delta, norm_delta, scaling, inv_norm, project_dist, escape_dist, eta1, eta2;
  int32_t max_iter;
  float tol, lam0;
  int32_t newton;
  int32_t max_blind;  // accepted steps below the rounding floor of F before giving up on further progress
  int32_t lds_frames, lds_terms;  // per-wave LDS rows: max frames / max terms over the model's components
  int32_t big_nh_rows;            // dexr_big_kernel: LDS rows reserved for the Hessian (n_max (n_max + 1) / 2)
  int32_t red_nj;                 // dexr_red_kernel: joints per component the LDS axis / origin rows are sized for
  float blind_tol;                // a Newton step of a verified, undamped model shorter than this is taken without a
                                  // further evaluation of the objective and ends the solve (0: off)
  float step_cap;                 // trust radius: a step whose largest component exceeds it is scaled down to it (0: off)
  float lam_jump;                 // on a rejected step lambda becomes at least lam_jump x mean diag(H) (0: plain x nu)
  float lam_fastdec;              // on an accepted step with rho > 0.9 lambda shrinks by this factor (0: Nielsen's 1/3)
  int32_t fastdec_max_rej;        // sixteen-lane kernel: ... only while the solve has had at most this many rejections
  float lam_recover;              // ... and by this factor while lambda is still above 10 lam0, i.e. while the damping
                                  // that a rejection raised is being taken back (0: lam_fastdec there too)
  float floor_scale;              // mixed-precision kernels: value differences below floor_scale x |F| are unverifiable
  int32_t stall_from;             // termination at the float rounding floor (see "stalled" in the kernels): from this
  float stall_ratio;              // many unverifiable ("blind") steps on, a step that is not < stall_ratio x the
  float stall_cap;                // previous one and is < stall_cap x tol ends the solve
  float* screen;                  // dexr_wide_kernel, screening launch: F(x0) per row (NULL: solve launch)
  float* screen_sum;              //   and its sum over the batch
  uint32_t q0;                    // queue mode: frames [0, q0) are handed out statically (wave w starts with tile w),
                                  // the queue counter numbers the frames from q0 on
  uint32_t qchunk;                // frames a wave takes from its component's queue per atomicAdd; 0 = tile mode
                                  // (wave w owns frames [64*tile, 64*tile+64), no queue traffic)
  int32_t n_kp;                   // keypoints per frame (21 for MediaPipe/MANO hands)
  int32_t h_origin[DEXR_MAXT];    // target_link_human_indices[0] per ref row (-1: position row = kp[h_task])
  int32_t h_task[DEXR_MAXT];      // target_link_human_indices[1] per ref row
  // ---- extended addressing (kernels instantiated with EXT = true; ignored otherwise) -------------------------------
  // Mixed-fleet batches: work item i of this launch is row perm[bucket[0] + i] of the batch arrays and the launch has
  // bucket[1] items -- both read from DEVICE memory, so the host never waits for the bucketing kernels.
  const int32_t* __restrict__ perm;    // NULL: item i is row i
  const int32_t* __restrict__ bucket;  // NULL: offset 0, B items
  int32_t ld;                          // row length of last / x0 / qout (>= n_opt; rows of a fleet batch are padded)
  int32_t ldf;                         // row length of fixed (>= n_fixed; = n_fixed outside fleet batches)
  // Frame sequences (SeqRetargeting.retarget semantics, seq_retarget.py:112-134): a work item is a SEQUENCE; the lane
  // (quad) that owns it solves its T frames in order and carries the raw solution, clipped to the joint limits
  // [lo + clip_eps, hi - clip_eps] (tables hold the optimiser's box, widened by clip_eps), as start point and
  // regularisation target of the next frame, and the DexPilot projection bits.  Frame t of sequence b reads input row
  // t * seq_stride + b and writes output row t * seq_stride + b; `last` and `state` have one row per sequence.
  int32_t T;                           // frames per sequence; 0 = independent frames (no clipping of `last`)
  int64_t seq_stride;                  // rows between consecutive frames of one sequence
  float clip_eps;
  // One-frame-per-wave launches of the sixteen-lane kernel (dexr_wide.hpp SPRINT): damping multiplier of each of the wave's four
  // rows -- {1, 1, 1, 1}: the rows are copies of one iteration; a ladder such as {0.03, 0.3, 3, 30}: every pass tries four
  // damping values from the accepted point and keeps the best acceptable trial point (dexr_tuning.sprint_ladder).
  float sprint_mu[4];
  int32_t iters_base;  // passes a frame has already had in an earlier launch (the tail launch of a large batch continues the count)
};

// Per-component side table of the sixteen-lanes-per-frame kernel (dexr_wide.hpp), derived from the component's table by
// the host when a model is created (dexr_api.hip: build_wide_tables): the kinematic tree cut into root-to-leaf chains,
// one per lane of a 16-lane row, and the revolute ancestors of every joint.
struct WideTable {
  int32_t n_chain, depth;       // root-to-leaf chains (<= 16) and the longest one (<= 16 joints)
  uint8_t chain[16][16];        // [lane][step]: local joint | 0x80 when this lane publishes it; 0xFF = none
  uint32_t anc_rev[DEXR_MAXJ];  // bit c set <=> joint c is a REVOLUTE ancestor-or-self of joint r
  // models with mimic joints (grid of the <= 16 optimised variables):
  uint8_t fam[16][4];           // joints that move with variable v (its own joint first), 0xFF-padded
  int32_t fam_max;              // largest family
  uint32_t pair[128];           // second-order joint pairs: k | j << 5 | local lower entry << 10 | doubled << 14,
  uint8_t pair_off[20];         // sorted by the lane that owns the target entry: lane l has [pair_off[l], pair_off[l+1])
  int32_t pair_max;             // longest per-lane list
};

// -DDEXR_SMALL_PROF=1 (tools/prof_small_stages.sh; never in the shipped library): wave 0 of the launch accumulates the cycles
// (s_memtime) of every stage of its passes and adds them to kp.g64out[stage] when its queue is dry.
#ifdef DEXR_SMALL_PROF
#define SPROF_DECL long long sp_t0 = 0, sp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SPROF_START() sp_t0 = clock64()
#define SPROF_STAGE(i)                    \
  {                                       \
    const long long sp_t1 = clock64();    \
    sp_acc[i] += sp_t1 - sp_t0;           \
    sp_t0 = sp_t1;                        \
  }
#define SPROF_FLUSH()                                                                   \
  if (wave_global == 0 && lane == 0 && kp.g64out) {                                     \
    for (int i = 0; i < 12; ++i) atomicAdd(&kp.g64out[i], (double)sp_acc[i]);           \
  }
#else
#define SPROF_DECL
#define SPROF_START()
#define SPROF_STAGE(i)
#define SPROF_FLUSH()
#endif

// -DDEXR_WAVE_TRACE=1 (tools/wave_trace.sh; never in the shipped library): every wave of a small-component solve launch
// records, on the 100 MHz wall clock (s_memrealtime: one time base for all XCDs), when it started, when its first frames
// were loaded, the end of each of its first 16 passes, when it retired, how many passes it ran and where it ran
// (HW_ID / XCC_ID): 24 doubles per wave in kp.g64out -- what a launch's duration is made of (dispatch ramp, first loads,
// passes under contention, tail).
#ifdef DEXR_WAVE_TRACE
#define WTRACE_DECL long long wt_t0 = wall_clock64(), wt_t1 = 0, wt_p[16]; int wt_n = 0;
#define WTRACE_LOADED() if (wt_t1 == 0) wt_t1 = wall_clock64();
#define WTRACE_PASS() { if (wt_n < 16) wt_p[wt_n] = wall_clock64(); ++wt_n; }
#define WTRACE_FLUSH()                                                                              \
  if (lane == 0 && kp.g64out) {                                                                     \
    double* o = kp.g64out + (size_t)wave_global * 24;                                               \
    o[0] = (double)wt_t0; o[1] = (double)wt_t1; o[2] = (double)wall_clock64(); o[3] = (double)wt_n; \
    o[4] = (double)__builtin_amdgcn_s_getreg(63492); o[5] = (double)__builtin_amdgcn_s_getreg(63508); \
    for (int i = 0; i < 16; ++i) o[8 + i] = i < wt_n ? (double)wt_p[i] : 0.0;                       \
  }
#else
#define WTRACE_DECL
#define WTRACE_LOADED()
#define WTRACE_PASS()
#define WTRACE_FLUSH()
#endif

enum { MODE_SOLVE = 0, MODE_EVAL = 1, MODE_FK = 2 };
enum { ST_CONVERGED = 0, ST_MAXITER = 1, ST_FALLBACK = 2 };  // == DEXR_STATUS_* in dexr.h

// FAST selects, for double, the hardware-estimate versions of 1 / sqrt, sqrt and division (see the specialisation below); float has one version.
template <typename real, bool FAST = false> struct RealTraits;
template <bool FAST> struct RealTraits<float, FAST> {
  static __device__ __forceinline__ float rsqrt(float v) { return __frsqrt_rn(v); }
  static __device__ __forceinline__ float sqrt(float v) { return __fsqrt_rn(v); }
  static __device__ __forceinline__ float div(float a, float b) { return a / b; }
  // sin/cos for joint angles (|a| <~ 1e3 rad): 2-term Cody-Waite reduction by pi/2 (the third term, 5.4e-15 k, is
  // below float32 resolution of the reduced argument here) + degree-7/8 minimax polynomials on [-pi/4, pi/4]; ~1 ulp,
  // no scratch, no slow path (ocml's sincosf carries a Payne-Hanek fallback that costs ~100 VGPRs and private memory
  // inside this kernel).
  static __device__ __forceinline__ void sincos(float a, float* s, float* c) {
    const float kf = rintf(a * 0.63661977236758134f);
    float r = fmaf(-kf, 1.5707962513e+00f, a);
    r = fmaf(-kf, 7.5497894159e-08f, r);
    const float z = r * r;
    const float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float cp = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const int k = (int)kf;
    const float ss = (k & 1) ? cp : sp;
    const float cc = (k & 1) ? sp : cp;
    *s = (k & 2) ? -ss : ss;
    *c = ((k + 1) & 2) ? -cc : cc;
  }
  static __device__ __forceinline__ float eps() { return 5.9604645e-8f; }
  // min(max(v, lo), hi) in one instruction (v_med3_f32; fmin / fmax of register operands cost a canonicalising v_max each)
  static __device__ __forceinline__ float clamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
};
// float64, correctly rounded: the validation / polish launches (dexr_retarget_f64 on the generic kernels is "the reference's own
// arithmetic": IEEE sqrt and division, rsqrt(inf) = 0, div(x, 0) = inf -- ADVICE r4).
template <> struct RealTraits<double, false> {
  static __device__ __forceinline__ double rsqrt(double v) { return 1.0 / ::sqrt(v); }
  static __device__ __forceinline__ double sqrt(double v) { return ::sqrt(v); }
  static __device__ __forceinline__ double div(double a, double b) { return a / b; }
  static __device__ __forceinline__ void sincos(double a, double* s, double* c) { ::sincos(a, s, c); }
  static __device__ __forceinline__ double eps() { return 1.1102230246251565e-16; }
  static __device__ __forceinline__ double clamp(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
};
// float64, FAST: the float64 tip pass (dexr_tip.hpp) and the general kernel (dexr_gen.hpp) only.
// 1 / sqrt(v), sqrt(v), a / b from the hardware estimates (v_rsq_f64 / v_rcp_f64, ~26 bits) + two Newton steps in ~10
// instructions, where a correctly rounded ::sqrt followed by an IEEE division is ~50 -- four times per pass in the Cholesky
// factorisation alone (round 4: the float64 tip pass spent a sixth of its instructions there).
// Accuracy: each Newton step squares the relative error (2^-26 -> 2^-51 -> rounding), the result is within 2 ulp of the
// correctly rounded value for normal, well-scaled arguments (pivots of a damped Hessian, residual norms, step lengths: all
// within 1e-30 .. 1e30 here).  NOT IEEE at the edges: rsqrt(inf) and div(x, 0) yield NaN (inf x 0 inside the Newton step)
// where the correctly rounded versions give 0 and inf, rsqrt(0) = NaN (callers test v > 0 first), denormal arguments lose
// accuracy.  The solver treats a NaN objective / step as a rejected step or a fallback to last_qpos, never as an answer.
template <> struct RealTraits<double, true> {
  static __device__ __forceinline__ double rsqrt(double v) {
    double r = __builtin_amdgcn_rsq(v);
    r = r * fma(-0.5 * v * r, r, 1.5);
    return r * fma(-0.5 * v * r, r, 1.5);
  }
  static __device__ __forceinline__ double sqrt(double v) { return v > 0.0 ? v * rsqrt(v) : 0.0; }
  static __device__ __forceinline__ double div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = r * fma(-b, r, 2.0);
    r = r * fma(-b, r, 2.0);
    return a * r;
  }
  static __device__ __forceinline__ void sincos(double a, double* s, double* c) { ::sincos(a, s, c); }
  static __device__ __forceinline__ double eps() { return 1.1102230246251565e-16; }
  static __device__ __forceinline__ double clamp(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
};

// Register-resident copy of one component's tables for small serial chains: every field is read with a compile-time
// index inside fully unrolled loops, so the values live in SGPRs (or SGPRs spilled to VGPR lanes) for the whole
// kernel instead of being re-fetched with s_load + s_waitcnt in every solver iteration (measured: 53 % of the wave
// cycles of the first persistent kernel were spent parked on those waits).
template <int NMAX>
struct LocalTab {
  static constexpr int LF = 4;  // frames kept locally
  static constexpr int LT = 4;  // terms kept locally
  int32_t n_joint, n_frame, n_term, n_base_frame;
  float X[NMAX][12];
  float lo[NMAX], hi[NMAX];
  int32_t api[NMAX], fbeg[NMAX], fend[NMAX];
  float frame_off[LF][3];
  int32_t term_task[LT], term_origin[LT], term_ref[LT];
  uint32_t term_mt[LT], term_mo[LT];  // ancestor masks of the task / origin frame of each term

  __device__ __forceinline__ void load(const dexr_comp_table& tb) {
    n_joint = tb.n_joint; n_frame = tb.n_frame; n_term = tb.n_term; n_base_frame = tb.n_base_frame;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
#pragma unroll
      for (int i = 0; i < 12; ++i) X[k][i] = tb.X[k][i];
      lo[k] = tb.lo[k]; hi[k] = tb.hi[k];
      api[k] = tb.api[k]; fbeg[k] = tb.fbeg[k]; fend[k] = tb.fend[k];
    }
#pragma unroll
    for (int f = 0; f < LF; ++f) {
#pragma unroll
      for (int i = 0; i < 3; ++i) frame_off[f][i] = tb.frame_off[f][i];
    }
#pragma unroll
    for (int t = 0; t < LT; ++t) {
      term_task[t] = tb.term_task[t]; term_origin[t] = tb.term_origin[t]; term_ref[t] = tb.term_ref[t];
      term_mt[t] = (t < tb.n_term) ? tb.frame_anc[tb.term_task[t]] : 0u;
      term_mo[t] = (t < tb.n_term && tb.term_origin[t] >= 0) ? tb.frame_anc[tb.term_origin[t]] : 0u;
    }
  }
};
}  // namespace dexr
#include "dexr_tip.hpp"
namespace dexr {
template <typename TB> struct TabTraits { static constexpr bool LOCAL = false; };
template <int N> struct TabTraits<LocalTab<N>> { static constexpr bool LOCAL = true; };

// CHAIN = true prunes, at compile time, everything a plain serial chain does not need: the component is one
// unbranched chain of exactly NMAX revolute joints hanging off the base, every joint is an optimised variable (no
// mimic / fixed-valued joints).  An Allegro or LEAP finger under VectorOptimizer is exactly that.
template <int NMAX, typename real, bool CHAIN = false, bool FASTM = false>
struct LaneSolver {
  static constexpr int NH = NMAX * (NMAX + 1) / 2;
  using RT = RealTraits<real, FASTM>;  // FASTM: the float64 tip kernel's estimate-based sqrt / division

  uint32_t revmask = ~0u;  // wave-uniform: bit k set = joint k is revolute (set by the kernel; saves the jtype loads)
  // ---- per-lane register state ---------------------------------------------------------------------------
  real x[NMAX];      // joint values of every local joint (optimised, fixed-valued and mimic alike)
  real xl[NMAX];     // regularisation target (last_qpos) for optimised joints
  real ax[NMAX][3];  // world joint axes
  real og[NMAX][3];  // world joint origins
  real g[NMAX];
  real H[NH];

  static __host__ __device__ __forceinline__ constexpr int hidx(int r, int c) { return r * (r + 1) / 2 + c; }

  template <typename T>
  static __device__ __forceinline__ T pick(const T (&arr)[NMAX], int idx) {
    // wave-uniform idx.  Written as a masked sum so that the optimiser cannot turn it back into a dynamically
    // indexed load (which would push the whole register-resident solver state into scratch memory).
    T v = 0;
#pragma unroll
    for (int s = 0; s < NMAX; ++s) v += (s == idx ? (T)1 : (T)0) * arr[s];
    return v;
  }

  // ---- forward kinematics of the component's joint list ---------------------------------------------------
  // Writes ax/og for every joint and the world position of every frame to LDS  P[(f*3+c)*64 + lane].
  template <typename TB>
  __device__ __forceinline__ void fk(const TB& tb, int nj, real* P, int lane) {
    constexpr bool LOCAL = TabTraits<TB>::LOCAL;  // LOCAL tables exist only for CHAIN kernels
    real R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    real p[3] = {0, 0, 0};
    real sR[DEXR_NSLOT][9];
    real sp[DEXR_NSLOT][3];
#pragma unroll
    for (int s = 0; s < DEXR_NSLOT; ++s) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sR[s][i] = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) sp[s][i] = 0;
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if (CHAIN || k < nj) {
        int rs;
        if constexpr (CHAIN) rs = (k == 0 ? -2 : -1);
        else rs = tb.restore[k];
        if (rs == -2) {
          R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
          p[0] = 0; p[1] = 0; p[2] = 0;
        } else if (rs >= 0) {
#pragma unroll
          for (int s = 0; s < DEXR_NSLOT; ++s)
            if (rs == s) {
#pragma unroll
              for (int i = 0; i < 9; ++i) R[i] = sR[s][i];
#pragma unroll
              for (int i = 0; i < 3; ++i) p[i] = sp[s][i];
            }
        }
        const float* X = tb.X[k];
        // p += R * Xp ; R = R * Xr
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] += R[3 * i] * (real)X[9] + R[3 * i + 1] * (real)X[10] + R[3 * i + 2] * (real)X[11];
        real Rn[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = R[3 * i] * (real)X[j] + R[3 * i + 1] * (real)X[3 + j] + R[3 * i + 2] * (real)X[6 + j];
        real q = x[k];
        bool revolute = true;
        if constexpr (!CHAIN) {
          if (tb.src_kind[k] == DEXR_SRC_MIMIC) {  // kinematics_adaptor.py:102-105
            q = (real)tb.mult[k] * pick(x, tb.src_idx[k]) + (real)tb.off[k];
            x[k] = q;
          }
          revolute = (revmask >> k) & 1u;
        }
        if (revolute) {
          real s, c;
          RT::sincos(q, &s, &c);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const real c0 = Rn[3 * i], c1 = Rn[3 * i + 1];
            R[3 * i] = c * c0 + s * c1;
            R[3 * i + 1] = c * c1 - s * c0;
            R[3 * i + 2] = Rn[3 * i + 2];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 9; ++i) R[i] = Rn[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) p[i] += q * Rn[3 * i + 2];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          ax[k][i] = R[3 * i + 2];
          og[k][i] = p[i];
        }
        int sv = -1;
        if constexpr (!CHAIN) sv = tb.save[k];
        if (sv >= 0) {
#pragma unroll
          for (int s = 0; s < DEXR_NSLOT; ++s)
            if (sv == s) {
#pragma unroll
              for (int i = 0; i < 9; ++i) sR[s][i] = R[i];
#pragma unroll
              for (int i = 0; i < 3; ++i) sp[s][i] = p[i];
            }
        }
        const int fb = tb.fbeg[k], fe = tb.fend[k];
        if constexpr (LOCAL) {
#pragma unroll
          for (int f = 0; f < TB::LF; ++f) {
            if (f >= fb && f < fe) {
              const real o0 = (real)tb.frame_off[f][0], o1 = (real)tb.frame_off[f][1], o2 = (real)tb.frame_off[f][2];
#pragma unroll
              for (int i = 0; i < 3; ++i)
                P[(f * 3 + i) * 64 + lane] = p[i] + R[3 * i] * o0 + R[3 * i + 1] * o1 + R[3 * i + 2] * o2;
            }
          }
        } else {
#pragma clang loop unroll(disable) vectorize(disable)
          for (int f = fb; f < fe; ++f) {
            const real o0 = (real)tb.frame_off[f][0], o1 = (real)tb.frame_off[f][1], o2 = (real)tb.frame_off[f][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
              P[(f * 3 + i) * 64 + lane] = p[i] + R[3 * i] * o0 + R[3 * i + 1] * o1 + R[3 * i + 2] * o2;
          }
        }
      }
    }
  }

  // ---- residuals (+ optional gradient / Hessian assembly) --------------------------------------------------
  // Returns the data term f(x) (reference "huber_distance", no regulariser).
  // ASM = 0: value only.  ASM = 1: value + g (data term only).  ASM = 2: value + g + H (data term only).
  template <int ASM, typename TB>
  __device__ __forceinline__ real residuals(const TB& tb, const KernelParams& kp, int nt, uint32_t vmask,
                                            const real* P, const real* T, const real* W, int lane) {
    constexpr bool LOCAL = TabTraits<TB>::LOCAL;
    const bool per_coord = (kp.kind == DEXR_KIND_POSITION);
    const bool weighted = (kp.kind == DEXR_KIND_DEXPILOT);
    const real beta = (real)kp.huber_delta;
    const real ibeta = (real)1 / beta;
    real F = 0;
    if (ASM >= 1) {
#pragma unroll
      for (int k = 0; k < NMAX; ++k) g[k] = 0;
    }
    if (ASM >= 2) {
#pragma unroll
      for (int i = 0; i < NH; ++i) H[i] = 0;
    }
    auto term = [&](int t, int ft, int fo, uint32_t mt_in, uint32_t mo_in) {
      real pt[3], po[3] = {0, 0, 0}, r[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) pt[i] = P[(ft * 3 + i) * 64 + lane];
      if (fo >= 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) po[i] = P[(fo * 3 + i) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) r[i] = pt[i] - po[i] - T[(t * 3 + i) * 64 + lane];
      real w = (real)kp.inv_norm;
      if (weighted) w *= W[t * 64 + lane];

      real fvec[3];  // dF/dr
      real hw[3];    // diagonal curvature weights of the loss in r
      real kap = 0;  // rank-one curvature correction: H -= kap * (J^T r)(J^T r)^T
      if (per_coord) {  // SmoothL1 per coordinate (optimizer.py:130,166)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const real e = r[i], ae = fabs(e);
          const bool quad = ae < beta;
          F += w * (quad ? (real)0.5 * e * e * ibeta : ae - (real)0.5 * beta);
          fvec[i] = w * (quad ? e * ibeta : (e > 0 ? (real)1 : (real)-1));
          // linear region: true curvature (0) when the second-order kinematic term is on (Newton), the IRLS
          // majoriser 1/|e| for Gauss-Newton (the majoriser costs ~3x the iterations on human targets)
          hw[i] = w * (quad ? ibeta : (kp.newton != 0 ? (real)0 : (real)1 / ae));
        }
      } else {  // SmoothL1 of the vector norm (optimizer.py:272-273, 534-541)
        const real d2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        const real d = RT::sqrt(d2);
        const bool quad = d < beta;
        F += w * (quad ? (real)0.5 * d2 * ibeta : d - (real)0.5 * beta);
        const real id = quad ? ibeta : (real)1 / d;  // d >= beta > 0 in the linear branch
        const real psi = w * id;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          fvec[i] = psi * r[i];
          hw[i] = psi;
        }
        kap = quad ? (real)0 : psi * id * id;
      }
      if (ASM >= 1) {
        const uint32_t mt = mt_in;
        const uint32_t mo = mo_in;
        const uint32_t mu = (mt | mo) & vmask;
        real col[NMAX][3];
        real u[NMAX];
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
          if ((mu >> k) & 1u) {
            const bool in_t = (mt >> k) & 1u, in_o = (mo >> k) & 1u;
            bool revolute = true;
            if constexpr (!CHAIN) revolute = (revmask >> k) & 1u;
            if (revolute) {
              real v[3] = {0, 0, 0};
              if (in_t) {
#pragma unroll
                for (int i = 0; i < 3; ++i) v[i] += pt[i] - og[k][i];
              }
              if (in_o) {
#pragma unroll
                for (int i = 0; i < 3; ++i) v[i] -= po[i] - og[k][i];
              }
              col[k][0] = ax[k][1] * v[2] - ax[k][2] * v[1];
              col[k][1] = ax[k][2] * v[0] - ax[k][0] * v[2];
              col[k][2] = ax[k][0] * v[1] - ax[k][1] * v[0];
            } else {
              const real sg = (in_t ? (real)1 : (real)0) - (in_o ? (real)1 : (real)0);
#pragma unroll
              for (int i = 0; i < 3; ++i) col[k][i] = sg * ax[k][i];
            }
            g[k] += col[k][0] * fvec[0] + col[k][1] * fvec[1] + col[k][2] * fvec[2];
            u[k] = col[k][0] * r[0] + col[k][1] * r[1] + col[k][2] * r[2];
          } else {
            col[k][0] = 0; col[k][1] = 0; col[k][2] = 0; u[k] = 0;
          }
        }
        if (ASM >= 2) {
          const bool newton = kp.newton != 0;
#pragma unroll
          for (int rr = 0; rr < NMAX; ++rr) {
            if ((mu >> rr) & 1u) {
              const real cw0 = hw[0] * col[rr][0], cw1 = hw[1] * col[rr][1], cw2 = hw[2] * col[rr][2];
              const real ku = kap * u[rr];
              // cf = col_r x f  (second-order term: d2p/dq_c dq_r . f = a_c . (col_r x f), c ancestor of r)
              const real cf0 = col[rr][1] * fvec[2] - col[rr][2] * fvec[1];
              const real cf1 = col[rr][2] * fvec[0] - col[rr][0] * fvec[2];
              const real cf2 = col[rr][0] * fvec[1] - col[rr][1] * fvec[0];
#pragma unroll
              for (int cc = 0; cc <= rr; ++cc) {
                if ((mu >> cc) & 1u) {
                  real h = cw0 * col[cc][0] + cw1 * col[cc][1] + cw2 * col[cc][2] - ku * u[cc];
                  const bool same = (((mt >> cc) & (mt >> rr)) | ((mo >> cc) & (mo >> rr))) & 1u;
                  bool rev_c = true;
                  if constexpr (!CHAIN) rev_c = (revmask >> cc) & 1u;
                  if (newton && same && rev_c)
                    h += ax[cc][0] * cf0 + ax[cc][1] * cf1 + ax[cc][2] * cf2;
                  H[hidx(rr, cc)] += h;
                }
              }
            }
          }
        }
      }
    };
    if constexpr (LOCAL) {
#pragma unroll
      for (int t = 0; t < TB::LT; ++t)
        if (t < nt) term(t, tb.term_task[t], tb.term_origin[t], tb.term_mt[t], tb.term_mo[t]);
    } else {
#pragma clang loop unroll(disable) vectorize(disable)
      for (int t = 0; t < nt; ++t) {
        const int ft = tb.term_task[t], fo = tb.term_origin[t];
        term(t, ft, fo, tb.frame_anc[ft], (fo >= 0) ? tb.frame_anc[fo] : 0u);
      }
    }
    return F;
  }

  // ---- mimic fold (kinematics_adaptor.py:107-113) applied to g (and H): x_k = m * x_s + b ------------------
  template <bool WITH_H, typename TB>
  __device__ __forceinline__ void fold_mimic(const TB& tb, int nj) {
    if constexpr (CHAIN) return;
    else {
    for (int k = 0; k < nj; ++k) {
      if (tb.src_kind[k] != DEXR_SRC_MIMIC) continue;
      const int s = tb.src_idx[k];
      const real m = (real)tb.mult[k];
      real gk = 0;
#pragma unroll
      for (int j = 0; j < NMAX; ++j)
        if (j == k) {
          gk = g[j];
          g[j] = 0;
        }
#pragma unroll
      for (int j = 0; j < NMAX; ++j)
        if (j == s) g[j] += m * gk;
      if (WITH_H) {
        real v[NMAX];  // v[j] = H[j][k]
#pragma unroll
        for (int j = 0; j < NMAX; ++j) v[j] = 0;
#pragma unroll
        for (int rr = 0; rr < NMAX; ++rr)
#pragma unroll
          for (int cc = 0; cc <= rr; ++cc) {
            if (cc == k) v[rr] = H[hidx(rr, cc)];
            else if (rr == k) v[cc] = H[hidx(rr, cc)];
          }
        const real hkk = pick(v, k);
        const real vs = pick(v, s);
#pragma unroll
        for (int rr = 0; rr < NMAX; ++rr)
#pragma unroll
          for (int cc = 0; cc <= rr; ++cc) {
            if (rr == k || cc == k) {
              H[hidx(rr, cc)] = 0;
            } else if (rr == s && cc == s) {
              H[hidx(rr, cc)] += (real)2 * m * vs + m * m * hkk;
            } else if (cc == s) {
              H[hidx(rr, cc)] += m * v[rr];
            } else if (rr == s) {
              H[hidx(rr, cc)] += m * v[cc];
            }
          }
      }
    }
    }
  }

  // ---- in-place Cholesky of H (lower) + solve H d = -g --------------------------------------------------------
  // Returns false where a pivot was not positive (indefinite Newton Hessian).  MODIFY = false: the step is then
  // meaningless and the caller raises the damping (one wasted pass).  MODIFY = true: modified Cholesky -- the offending
  // pivot is replaced by max(|pivot|, pivot_floor), i.e. negative curvature along that pivot direction is reflected,
  // and d is a descent direction of the model H + E: the pass is not wasted (far starts spent 4-7 of their 15 passes on
  // failed factorisations, tools/lm_lab.py); the caller must not treat such a step as a verified Newton step.
  template <bool MODIFY = false>
  __device__ __forceinline__ bool chol_solve(real (&d)[NMAX], real pivot_floor = 1) {
    bool ok = true;
    real inv[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
      real dj = H[hidx(j, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) dj -= H[hidx(j, k)] * H[hidx(j, k)];
      if (!(dj > (MODIFY ? (real)1e-6 * pivot_floor : (real)1e-30))) {
        ok = false;
        dj = MODIFY ? fmax(fabs(dj), pivot_floor) : (real)1;
      }
      const real iv = RT::rsqrt(dj);
      inv[j] = iv;
      H[hidx(j, j)] = dj * iv;
#pragma unroll
      for (int i = j + 1; i < NMAX; ++i) {
        real s = H[hidx(i, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= H[hidx(i, k)] * H[hidx(j, k)];
        H[hidx(i, j)] = s * iv;
      }
    }
    // forward: L y = -g
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      real s = -g[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= H[hidx(i, k)] * d[k];
      d[i] = s * inv[i];
    }
    // backward: L^T d = y
#pragma unroll
    for (int i = NMAX - 1; i >= 0; --i) {
      real s = d[i];
#pragma unroll
      for (int k = i + 1; k < NMAX; ++k) s -= H[hidx(k, i)] * d[k];
      d[i] = s * inv[i];
    }
    return ok;
  }
};

// ------------------------------------------------------------------------------------------------------------
// Kernel: one wave = 64 items x one component.  blockDim.x = 64 * waves_per_block; dynamic LDS =
// waves_per_block * 64 * sizeof(real) * (3*lds_frames + 4*lds_terms).
// EXT = true adds the extended addressing of KernelParams (fleet buckets, frame sequences).  The small-component
// kernels are instantiated both ways so that the plain single-model launch keeps its register budget; the large ones
// always carry it.
// TIP = true (CHAIN, 4 joints, solve only): every component is a tip component and its pass is dexr_tip.hpp's (float32:
// packed arithmetic, constants pinned; float64 since round 4: the reference's own arithmetic type on the same pass).
template <int NMAX, typename real, int MODE, bool CHAIN = false, bool EXT = (NMAX > 8), bool TIP = false>
__global__ void __launch_bounds__((TIP && sizeof(real) == 4) ? DEXR_TIP_BLOCK_MAX : DEXR_BLOCK_MAX, (CHAIN && NMAX <= 4 && sizeof(real) == 4) ? DEXR_CHAIN_MINW : (TIP ? DEXR_TIP64_MINW : 1)) dexr_kernel(const KernelParams kp, const dexr_comp_table* __restrict__ comps) {
  static_assert(!TIP || (CHAIN && NMAX == 4 && MODE == MODE_SOLVE), "tip pass: 4-joint chain solve only");
  extern __shared__ __align__(16) unsigned char lds_raw[];
  using LS = LaneSolver<NMAX, real, CHAIN, TIP>;
  using RT = RealTraits<real, TIP>;
  const int lane = threadIdx.x & 63;
  // readfirstlane: tell the compiler the wave index (hence the component, every table address and every branch on
  // table contents) is wave-uniform -> s_load / SGPR operands / scalar branches instead of per-lane loads + exec masks
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * waves_per_block + wave_in_block;
  const int comp = (int)(wave_global % kp.n_comp);
  const int64_t tile = wave_global / kp.n_comp;
  constexpr bool PERSISTENT = (MODE == MODE_SOLVE && NMAX <= 8);  // lanes pull frames from a queue (see below)
  // number of work items and their rows (see KernelParams: fleet buckets are sized and listed on the device)
  const int64_t nB = (EXT && kp.bucket) ? (int64_t)kp.bucket[1] : kp.B;
  const int64_t pbase = (EXT && kp.bucket) ? (int64_t)kp.bucket[0] : 0;
  auto row_of = [&](int64_t it) -> int64_t { return (EXT && kp.perm) ? (int64_t)kp.perm[pbase + it] : it; };
  const int ld = EXT ? kp.ld : kp.n_opt;
  const bool seq = EXT && MODE == MODE_SOLVE && kp.T > 0;
  if (nB <= 0) return;
  if (!PERSISTENT && tile * 64 >= nB) return;
  const int64_t item_raw = tile * 64 + lane;
  const bool valid = item_raw < nB;
  const int64_t item = valid ? item_raw : nB - 1;

  // PARK (round 6, float64 tip kernel): the accepted point's model -- Hessian (10), gradient (4) -- and the point the pass
  // stepped from (4) live in per-lane LDS slots ([value][lane], 18 doubles = 9 KB per wave) instead of 36 VGPRs: at 256 VGPRs
  // the compiler had 37 registers of exactly this cross-pass state in scratch (128 B per lane, WRITE_SIZE 35.8 MB per launch
  // for 4.2 MB of output, profiles/pmc_allegro_vector_f64.json); each value is read and written once per pass.
  constexpr bool PARK = TIP && sizeof(real) == 8;
  constexpr int NPARK = PARK ? (NMAX * (NMAX + 1) / 2 + 2 * NMAX) : 0;
  const int per_wave = 64 * (3 * kp.lds_frames + 4 * kp.lds_terms + (TIP ? 1 : 0) + NPARK);  // (+ the tip pass's broadcast constants)
  real* P = reinterpret_cast<real*>(lds_raw) + (size_t)wave_in_block * per_wave;
  real* T = P + 64 * 3 * kp.lds_frames;
  real* W = T + 64 * 3 * kp.lds_terms;
  real* PK = W + 64 * kp.lds_terms + 64 + lane;  // PARK: value i of this lane at PK[64 * i]

  // `comps` is a separate __restrict__ kernel argument (not a struct member) so that the compiler may treat the
  // tables as invariant and read them with scalar loads
  const dexr_comp_table& tb = comps[comp];
  const int nj = CHAIN ? NMAX : tb.n_joint, nt = tb.n_term;

  LS S;
  uint32_t vmask = 0;   // joints that carry an optimisation variable (directly or as a mimic)
  if constexpr (!CHAIN) {
    uint32_t rm = 0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if (k < nj && tb.jtype[k] == DEXR_JOINT_REVOLUTE) rm |= 1u << k;
    S.revmask = rm;
  }
  uint32_t optmask = 0; // joints that ARE an optimisation variable
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (CHAIN || k < nj) {
      const int sk = CHAIN ? DEXR_SRC_OPT : tb.src_kind[k];
      if (sk == DEXR_SRC_OPT) {
        vmask |= 1u << k;
        optmask |= 1u << k;
      } else if (sk == DEXR_SRC_MIMIC) {
        vmask |= 1u << k;
      }
    }
  }

  // one row of ref_value for frame `it`: either handed in directly or formed from raw hand keypoints as the callers
  // of the reference do (joint_pos[task] - joint_pos[origin] / joint_pos[idx], profile_online_retargeting.py:24-30)
  auto ref_row = [&](int64_t it, int row, float (&rv)[3]) {  // `it`: row of the input arrays
    if (kp.kpts) {
      const float* a = kp.kpts + (it * kp.n_kp + kp.h_task[row]) * 3;
      const int o = kp.h_origin[row];
      if (o >= 0) {
        const float* b = kp.kpts + (it * kp.n_kp + o) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = a[i] - b[i];
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = a[i];
      }
    } else {
      const float* r = kp.ref + (it * kp.n_ref + row) * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) rv[i] = r[i];
    }
  };

  // column of last_qpos / qpos_out of local joint k.  A tip component's four joints are consecutive columns (checked when the
  // kernel is selected): one pinned base index instead of four table loads, and the per-joint addresses share one base
  constexpr bool TIP32 = TIP && sizeof(real) == 4;  // the float32 tip kernel prunes every path only float64 launches take
  const int tip_api0 = TIP ? tip_pin((int)tb.api[0]) : 0;
  auto api_of = [&](int k) -> int { if constexpr (TIP) return tip_api0 + k; else return tb.api[k]; };

  // Loads everything frame `it` needs into this lane: joint values (start point, regularisation target, fixed
  // joints) into registers, per-term targets / DexPilot weights into the lane's LDS column.  Returns the updated
  // DexPilot projection bits.
  // `item`: work item (frame, or sequence in sequence mode); t > 0 (sequence mode only): frame t of the sequence, whose
  // start point / regularisation target is the lane's own previous solution (carry) and whose incoming DexPilot bits
  // are `st_carry`.
  auto load_item = [&](int64_t item_i, int t = 0, uint32_t st_carry = 0u) -> uint32_t {
    const int64_t r0 = row_of(item_i);                       // row of `last` / `state` (one per item)
    const int64_t it = seq ? (int64_t)t * kp.seq_stride + r0 : r0;  // row of this frame's inputs
    const bool carry = seq && t > 0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      real prev = S.x[k];
      S.x[k] = 0;
      S.xl[k] = 0;
      if (CHAIN || k < nj) {
        const int sk = CHAIN ? DEXR_SRC_OPT : tb.src_kind[k];
        if (sk == DEXR_SRC_OPT) {
          real v;
          if (MODE == MODE_EVAL) v = (real)kp.xin[it * kp.n_opt + api_of(k)];
          else if (carry) v = (real)(float)prev;  // the reference carries the float32 result (optimizer.py:99)
          else if (!TIP32 && kp.x0) v = (real)kp.x0[r0 * ld + api_of(k)];  // (float64 polish launches only)
          else v = (real)kp.last[r0 * ld + api_of(k)];
          real l = carry ? v : (real)kp.last[r0 * ld + api_of(k)];
          if (seq) {  // seq_retarget.py:118-120: last_qpos clipped to the joint limits before every solve
            const real lo_s = (real)tb.lo[k] + (real)kp.clip_eps, hi_s = (real)tb.hi[k] - (real)kp.clip_eps;
            l = fmin(fmax(l, lo_s), hi_s);
            v = l;
          }
          S.xl[k] = l;
          S.x[k] = v;
        } else if (sk == DEXR_SRC_FIXED) {
          S.x[k] = (real)tb.mult[k] * (real)kp.fixed[it * kp.ldf + tb.src_idx[k]] + (real)tb.off[k];
        } else if (sk == DEXR_SRC_DIRECT) {
          S.x[k] = (real)kp.xin[it * kp.n_q + tb.src_idx[k]];
        }
      }
    }
    if (MODE == MODE_FK) return 0u;
    uint32_t nst = 0;
    if (!TIP && kp.kind == DEXR_KIND_DEXPILOT) {
      // optimizer.py:462-508.  Terms are the model's vectors in order: pairs first, then wrist->finger.
      const int F = kp.num_fingers;
      const int n_pair = F * (F - 1) / 2, len_s1 = F - 1;
      const uint32_t st = carry ? st_carry : (kp.state ? kp.state[r0] : 0u);
      for (int i = 0; i < len_s1; ++i) {
        float rv[3];
        ref_row(it, i, rv);
        const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
        bool b = (st >> i) & 1u;
        if (dist < kp.project_dist) b = true;
        if (dist > kp.escape_dist) b = false;
        nst |= (b ? 1u : 0u) << i;
      }
      int idx = len_s1;
      for (int a = 0; a < F - 2; ++a)
        for (int b2 = a + 1; b2 < F - 1; ++b2) {
          float rv[3];
          ref_row(it, idx, rv);
          const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
          const bool b = ((nst >> b2) & 1u) && ((nst >> a) & 1u) && (dist <= 0.03f);
          nst |= (b ? 1u : 0u) << idx;
          ++idx;
        }
      for (int t = 0; t < nt; ++t) {
        const int row = tb.term_ref[t];
        float rv[3];
        ref_row(it, row, rv);
        float tv[3];
        float wt;
        if (row < n_pair) {
          const bool pr = (nst >> row) & 1u;
          if (pr) {
            const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
            const float eta = row < len_s1 ? kp.eta1 : kp.eta2;
#pragma unroll
            for (int i = 0; i < 3; ++i) tv[i] = (rv[i] / (dist + 1e-6f)) * eta;
            wt = row < len_s1 ? 200.f : 400.f;
          } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) tv[i] = rv[i] * kp.scaling;
            wt = 1.f;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) tv[i] = rv[i] * kp.scaling;
          wt = (float)(n_pair + F);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) T[(t * 3 + i) * 64 + lane] = (real)tv[i];
        W[t * 64 + lane] = (real)wt;
      }
    } else {
      const float sc = (TIP || kp.kind == DEXR_KIND_VECTOR) ? kp.scaling : 1.f;
#pragma clang loop unroll(disable) vectorize(disable)
      for (int t = 0; t < nt; ++t) {
        float rv[3];
        ref_row(it, tb.term_ref[t], rv);
#pragma unroll
        for (int i = 0; i < 3; ++i) T[(t * 3 + i) * 64 + lane] = (real)(rv[i] * sc);  // f32 multiply: optimizer.py:246
      }
    }
    return nst;
  };

  // base frames never move
#pragma clang loop unroll(disable) vectorize(disable)
  for (int f = 0; f < tb.n_base_frame; ++f) {
#pragma unroll
    for (int i = 0; i < 3; ++i) P[(f * 3 + i) * 64 + lane] = (real)tb.frame_off[f][i];
  }

  uint32_t nst_out = 0;  // written back at the very end: no global store may precede the (invariant) table loads
  if (!PERSISTENT) nst_out = load_item(item);

  if (MODE == MODE_FK) {
    S.fk(tb, nj, P, lane);
    if (valid) {
      for (int t = 0; t < nt; ++t) {
        const int f = tb.term_task[t], row = tb.term_ref[t];
#pragma unroll
        for (int i = 0; i < 3; ++i) kp.f64out[(row_of(item) * kp.n_ref + row) * 3 + i] = (double)P[(f * 3 + i) * 64 + lane];
      }
    }
    return;
  }

  const real delta = (real)kp.norm_delta;

  if (MODE == MODE_EVAL) {  // objective(x, grad): value without, gradient with the regulariser (quirk Q1)
    S.fk(tb, nj, P, lane);
    const real f = S.template residuals<1>(tb, kp, nt, vmask, P, T, W, lane);
    S.template fold_mimic<false>(tb, nj);
    if (valid) {
      const int64_t r0 = row_of(item);
      if (kp.kind == DEXR_KIND_DEXPILOT && kp.state && comp == 0) kp.state[r0] = nst_out;
      atomicAdd(&kp.f64out[r0], (double)f);
#pragma unroll
      for (int k = 0; k < NMAX; ++k)
        if ((optmask >> k) & 1u)
          kp.g64out[r0 * kp.n_opt + tb.api[k]] = (double)(S.g[k] + (real)2 * delta * (S.x[k] - S.xl[k]));
    }
    return;
  }

  // ---- MODE_SOLVE: projected Levenberg-Marquardt / Newton --------------------------------------------------
  if constexpr (NMAX <= 8) {
    // ---- small components: PERSISTENT LANES.  Frames need different iteration counts (warm starts: 3-5, a few
    // need 15+), so a wave that owns a fixed tile of 64 frames idles most lanes most of the time.  Here a lane that
    // finishes a frame immediately pulls the next one from a per-component work queue (wave-local pool refilled by
    // one atomicAdd per QCHUNK frames), so every lane does useful iterations until the queue is dry.
    // Per iteration: ONE forward-kinematics pass + ONE fused value/gradient/Hessian assembly at the trial point +
    // one Cholesky.  The accepted quadratic model (Hs, gs) stays in registers, so a rejected step costs only the
    // re-solve with more damping.
    auto run = [&](const auto& tbl) {
    // loop-invariant scalars of the pass; the tip kernel pins them in SGPRs (a kernel-argument load + wait each otherwise)
    constexpr bool PIN = TIP && sizeof(real) == 4;  // (the float64 tip kernel does not pin: SGPR pairs run out)
    auto hot = [](float v) -> float { if constexpr (PIN) return tip_pin(v); else return v; };
    auto hoti = [](int v) -> int { if constexpr (PIN) return tip_pin(v); else return v; };
    const real k_lam0 = (real)hot(kp.lam0), k_tol = (real)hot(kp.tol), k_blind_tol = (real)hot(kp.blind_tol);
    const real k_step_cap = (real)hot(kp.step_cap), k_lam_jump = (real)hot(kp.lam_jump), k_lam_fastdec = (real)hot(kp.lam_fastdec);
    const real k_lam_recover = (real)hot(kp.lam_recover), k_stall_ratio = (real)hot(kp.stall_ratio), k_stall_cap = (real)hot(kp.stall_cap);
    const int k_stall_from = hoti(kp.stall_from), k_max_blind = hoti(kp.max_blind), k_max_iter = hoti(kp.max_iter);
    const real k_delta = (real)hot((float)delta);
    // (tip kernel) loss constants of the component's single term
    const real tip_beta = (real)hot(kp.huber_delta), tip_ibeta = sizeof(real) == 4 ? (real)hot(1.f / kp.huber_delta) : (real)1 / (real)kp.huber_delta;
    const real tip_w = (real)hot(kp.inv_norm), tip_nw = (real)hot(kp.newton != 0 ? 1.f : 0.f);
    const unsigned QCHUNK = kp.qchunk;
    real Hs[PARK ? 1 : LS::NH], gs[PARK ? 1 : NMAX], xo[PARK ? 1 : NMAX];  // (PARK: in LDS, see PK)
    auto Hs_get = [&](int i) -> real { if constexpr (PARK) return PK[64 * i]; else return Hs[i]; };
    auto gs_get = [&](int k) -> real { if constexpr (PARK) return PK[64 * (LS::NH + k)]; else return gs[k]; };
    real F = 0, lam = k_lam0, nu = 2, sprev = (real)1e30;
    int my_iters = 0, blind = 0, nrej = 0;  // nrej: rejected steps of this solve (fast damping recovery only after the first)
    bool has = false, fresh = false;
    int64_t my_item = 0;
    int my_t = 0;  // sequence mode: frame of the lane's sequence being solved
    uint32_t my_nst = 0;
    unsigned pool_next = 0, pool_end = 0;  // wave-uniform
    bool dry = false;                      // wave-uniform: the queue has been exhausted
    if (QCHUNK == 0) {  // tile mode: this wave owns frames [64*tile, 64*tile+64) and never touches the queue
      pool_next = (unsigned)(tile * 64);
      pool_end = (unsigned)((tile * 64 + 64 < nB) ? tile * 64 + 64 : nB);
      if ((int64_t)pool_next >= nB) return;
    } else {  // queue mode: the first tile is static as well (no start-up stampede on the counter).  The grid is rounded
      // up to whole blocks: a surplus wave (tile * 64 >= q0, where the queue's numbering starts) gets no static tile.
      const bool in_static = tile * 64 < (int64_t)kp.q0 && tile * 64 < nB;
      pool_next = in_static ? (unsigned)(tile * 64) : 0u;
      pool_end = in_static ? (unsigned)((tile * 64 + 64 < nB) ? tile * 64 + 64 : nB) : 0u;
    }
    unsigned* queue = kp.queue + comp;
    SPROF_DECL
    WTRACE_DECL

    for (;;) {
      SPROF_START();
      // (1) hand new frames to idle lanes
      const unsigned long long want = __ballot(!has);
      if (want != 0ull && !dry) {
        if (pool_next >= pool_end && QCHUNK == 0) dry = true;
        if (pool_next >= pool_end && !dry) {
          unsigned base = 0;
          if (lane == 0) base = atomicAdd(queue, QCHUNK);
          base = __builtin_amdgcn_readfirstlane(base) + kp.q0;
          if ((int64_t)base >= nB) {
            dry = true;
          } else {
            pool_next = base;
            pool_end = (unsigned)(((int64_t)base + QCHUNK < nB) ? base + QCHUNK : nB);
          }
        }
        if (!dry) {
          const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
          const unsigned cand = pool_next + rank;
          const bool got = !has && cand < pool_end;
          const unsigned taken = (unsigned)__popcll(__ballot(got));
          pool_next += taken;
          if (got) {
            my_item = (int64_t)cand;
            my_t = 0;
            my_nst = load_item(my_item);
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
              if ((optmask >> k) & 1u) S.x[k] = RT::clamp(S.x[k], (real)tbl.lo[k], (real)tbl.hi[k]);
            has = true;
            fresh = true;
            lam = k_lam0;
            nu = 2;
            sprev = (real)1e30;
            my_iters = 0;
            blind = 0;
            nrej = 0;
          }
        }
      }
      if (!__any(has)) {
        if (dry) break;
        continue;
      }
      SPROF_STAGE(0)
      WTRACE_LOADED()

      // (2) step from the accepted model (lanes holding a fresh frame evaluate their start point instead)
      real smax = 0, pred = 0;
      bool ok = true, last_step = false;
      if (__any(has && !fresh)) {
        uint32_t freemask = 0;
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
          const bool isopt = (optmask >> k) & 1u;
          // (bitwise, not short-circuit: `&&` / `||` on lane-varying conditions compile to nested exec-mask branches -- 18
          // scalar / branch instructions per joint here -- where four compares and three mask operations do)
          const real gk = gs_get(k);
          const bool act = (bool)(((int)(S.x[k] <= (real)tbl.lo[k]) & (int)(gk > 0)) | ((int)(S.x[k] >= (real)tbl.hi[k]) & (int)(gk < 0)));
          if (isopt && !act) freemask |= 1u << k;
          S.g[k] = (isopt && !act) ? gk : (real)0;
        }
#pragma unroll
        for (int rr = 0; rr < NMAX; ++rr) {
          const bool fr = (freemask >> rr) & 1u;
#pragma unroll
          for (int cc = 0; cc < rr; ++cc) {
            const bool fc = (freemask >> cc) & 1u;
            S.H[LS::hidx(rr, cc)] = (fr && fc) ? Hs_get(LS::hidx(rr, cc)) : (real)0;
          }
          S.H[LS::hidx(rr, rr)] = fr ? Hs_get(LS::hidx(rr, rr)) + (real)2 * k_delta + lam : (real)1;
        }
        real gm[NMAX];
#pragma unroll
        for (int k = 0; k < NMAX; ++k) gm[k] = S.g[k];
        real d[NMAX];
        // modified Cholesky (see chol_solve): `ok` = no pivot had to be modified, i.e. d is the Newton step of the
        // damped model; a modified step is still tried (the decrease test below judges it)
        ok = S.template chol_solve<true>(d, (real)2 * k_delta + lam);
        const bool stepping = has && !fresh;
        // trust radius: scale the step so that no joint moves more than step_cap (alpha in (0, 1])
        real dmax = 0, gd = 0, dd = 0;
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if ((optmask >> k) & 1u) {
            dmax = fmax(dmax, fabs(d[k]));
            gd -= gm[k] * d[k];
            dd += d[k] * d[k];
          }
        // (a step from a modified factorisation -- negative curvature -- is stretched (up to 8 x) towards the trust radius, see dexr_red.hpp)
        const bool cut = (bool)((int)(k_step_cap > 0) & ((int)(dmax > k_step_cap) | ((int)!ok & (int)(dmax > (real)0))));
        const real alpha = cut ? fmin(RT::div(k_step_cap, dmax), (real)8) : (real)1;
        // predicted decrease of the damped model along alpha*d:  alpha (1 - alpha/2) (-g.d) + alpha^2/2 lam d.d
        pred = alpha * ((real)1 - (real)0.5 * alpha) * gd + (real)0.5 * alpha * alpha * lam * dd;
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
          if constexpr (PARK) PK[64 * (LS::NH + NMAX + k)] = S.x[k]; else xo[k] = S.x[k];
          if ((optmask >> k) & 1u) {
            const real xt = RT::clamp(S.x[k] + alpha * d[k], (real)tbl.lo[k], (real)tbl.hi[k]);
            smax = fmax(smax, fabs(xt - S.x[k]));
            if (stepping) S.x[k] = xt;
          }
        }
        // The model at the accepted point is verified and (lambda back at its floor) undamped: a Newton step this short
        // is the converged answer to well below the tolerance, evaluating the objective there once more could only
        // confirm it.  Take it and retire the frame one pass earlier.
        // Beyond 10 tol this is only trusted on the fast (quadratic) tail of a Newton iteration: the step must be at most a
        // tenth of the previous accepted one -- in a nearly flat valley steps shrink slowly and C s^2 is not small.
        last_step = (bool)((int)stepping & (int)ok & (int)(smax < k_blind_tol) & (int)(lam <= k_lam0) &
                           ((int)(smax < (real)10 * k_tol) | (int)(smax < (real)0.1 * sprev)));
      }

      // (3) forward kinematics + fused value / gradient / Hessian at S.x -- unless every lane that still holds a frame is
      // taking its blind last step: those lanes retire on the step alone (the objective there "could only confirm it"), so the
      // wave's last pass, typically, needs no evaluation at all (skipped only when the caller did not ask for the final
      // objective values, which would otherwise be those of the point before)
      SPROF_STAGE(1)
      real Ft = F;
      const bool need_eval = kp.fval != nullptr || __any(has && !last_step);  // wave-uniform
      if (need_eval) {
        if constexpr (TIP) {
          Ft = tip_eval<real>(tbl, S.x, T[lane], T[64 + lane], T[128 + lane], tip_beta, tip_ibeta, tip_w, tip_nw, S.g, S.H);
          SPROF_STAGE(2)
        } else {
          S.fk(tb, nj, P, lane);
          SPROF_STAGE(2)
          Ft = S.template residuals<2>(tb, kp, nt, vmask, P, T, W, lane);
          S.template fold_mimic<true>(tb, nj);
        }
        SPROF_STAGE(3)
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
          if ((optmask >> k) & 1u) {
            const real dx = S.x[k] - S.xl[k];
            Ft += k_delta * dx * dx;
            S.g[k] += (real)2 * k_delta * dx;
          } else {
            S.g[k] = 0;
          }
        }
      }

      // (4) accept / reject, damping update, termination
      bool accept = false, finished = false;
      int status = ST_MAXITER;
      if (has) {
        const bool finite = (bool)((int)(Ft == Ft) & (int)(smax == smax) & (int)(fabs(Ft) < (real)1e30));
        if (fresh) {
          accept = true;  // start point: adopt its model unconditionally
          fresh = false;
          F = Ft;
          if (!finite) {
            finished = true;
            status = ST_FALLBACK;
          }
        } else if (last_step && finite) {
          accept = true;
          finished = true;
          status = ST_CONVERGED;
          ++my_iters;
          F = Ft;
        } else {
          // resolution of F in this arithmetic: rounding of the sum (16 eps |F|) and, when F itself is small, of the frame
          // positions behind it (1e-8 m x a force of ~0.1: ~2e-9 in float32 -- a Newton step whose predicted decrease is
          // below that cannot be verified, only trusted)
          const real noise = (real)16 * RT::eps() * fmax(fabs(F), (real)2e-3);
          // below the rounding floor of F the decrease test is meaningless: trust the (small) Newton step
          const bool below_floor = (bool)((int)ok & (int)finite & (int)(pred <= noise) & (int)(smax < (real)1e-2));
          accept = (bool)((int)finite & ((int)(Ft <= F) | (int)below_floor));  // (a modified-Cholesky step is judged by the decrease alone)
          ++my_iters;
          if (accept) {
            const real rho = RT::div(F - Ft, fmax(pred, (real)1e-30));
            const real t = (real)2 * rho - (real)1;
            real shrink = below_floor ? (real)(1.0 / 3.0) : fmax((real)(1.0 / 3.0), (real)1 - t * t * t);
            if ((bool)((int)(k_lam_fastdec > 0) & (int)(rho > (real)0.9)))
              shrink = (bool)((int)(k_lam_recover > 0) & (int)(nrej <= 2) & (int)(lam > (real)10 * k_lam0)) ? k_lam_recover : k_lam_fastdec;
            lam = fmax(lam * shrink, (real)1e-9);
            nu = 2;
            F = Ft;
            const bool stalled = (bool)((int)below_floor & (int)(blind >= k_stall_from) & (int)(smax > k_stall_ratio * sprev) & (int)(smax < k_stall_cap * k_tol));
            blind = below_floor ? blind + 1 : 0;
            sprev = smax;
            // a step below tol only means convergence when the damping is not what made it small (see dexr_big.hpp)
            const real lam_ok = fmax((real)2 * k_delta, (real)10 * k_lam0);
            if ((bool)(((int)(smax < k_tol) & (int)(lam <= lam_ok)) | (int)stalled | (int)(blind >= k_max_blind))) {
              finished = true;
              status = ST_CONVERGED;
            } else if (smax < k_tol) {
              lam = fmax((real)0.1 * lam, (real)0.5 * lam_ok);
            }
          } else {
            ++nrej;
        lam = fmax(lam, (real)1e-6) * nu;
            if (k_lam_jump > 0) {  // go straight to a damping that matters next to the curvature
              real ds = 0;
              int dn = 0;
#pragma unroll
              for (int k = 0; k < NMAX; ++k)
                if ((optmask >> k) & 1u) {
                  ds += Hs_get(LS::hidx(k, k));
                  ++dn;
                }
              lam = fmax(lam, k_lam_jump * ds / (real)(dn > 0 ? dn : 1));
            }
            nu *= 2;
            if (lam > (real)1e10) {  // no descent direction resolvable any more
              finished = true;
              status = finite ? ST_CONVERGED : ST_FALLBACK;
            }
            if ((bool)((int)finite & (int)(smax < k_tol))) {  // rejected step below tol: converged at the rounding floor of F
              finished = true;
              status = ST_CONVERGED;
            }
          }
          finished = (bool)((int)finished | (int)(my_iters >= k_max_iter));  // status stays ST_MAXITER
        }
      }
      WTRACE_PASS()
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if constexpr (PARK) {
          // (a lane that stepped reads back the point it stepped from; `fresh` lanes -- accept is true for them -- never wrote one)
          if (!accept) S.x[k] = PK[64 * (LS::NH + NMAX + k)];
          if (accept) PK[64 * (LS::NH + k)] = S.g[k];
        } else {
          S.x[k] = accept ? S.x[k] : xo[k];
          gs[k] = accept ? S.g[k] : gs[k];
        }
      }
      if constexpr (PARK) {
        if (accept) {
#pragma unroll
          for (int i = 0; i < LS::NH; ++i) PK[64 * i] = S.H[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < LS::NH; ++i) Hs[i] = accept ? S.H[i] : Hs[i];
      }

      // (5) retire finished frames
      SPROF_STAGE(4)
#ifdef DEXR_SMALL_PROF
      sp_acc[11] += 1;
#endif
      if (finished) {
        bool bad = (status == ST_FALLBACK);
#pragma unroll
        for (int k = 0; k < NMAX; ++k)
          if ((optmask >> k) & 1u) bad = bad || !(S.x[k] == S.x[k]);
        if (bad) status = ST_FALLBACK;
        const int64_t r0 = row_of(my_item);
        const int64_t orow = seq ? (int64_t)my_t * kp.seq_stride + r0 : r0;  // output row of this frame
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
          if ((optmask >> k) & 1u) {
            const real v = bad ? S.xl[k] : S.x[k];
            S.x[k] = v;  // (sequence mode: the value the next frame starts from)
            kp.qout[orow * ld + api_of(k)] = (float)v;
            if (!TIP32 && kp.qout64) kp.qout64[orow * ld + api_of(k)] = (double)v;  // (float64 launches only)
          }
        }
        if (kp.status) atomicMax(&kp.status[orow], status);
        if (kp.iters) atomicMax(&kp.iters[orow], my_iters);
        if (kp.fval) atomicAdd(&kp.fval[orow], (float)F);
        if (seq && my_t + 1 < kp.T) {
          // next frame of the same sequence: carry the solution (clipped inside load_item) and the projection bits
          ++my_t;
          my_nst = load_item(my_item, my_t, my_nst);
#pragma unroll
          for (int k = 0; k < NMAX; ++k)
            if ((optmask >> k) & 1u) S.x[k] = RT::clamp(S.x[k], (real)tbl.lo[k], (real)tbl.hi[k]);
          fresh = true;
          lam = k_lam0;
          nu = 2;
          sprev = (real)1e30;
          my_iters = 0;
          blind = 0;
          nrej = 0;
        } else {
          if (!TIP && kp.kind == DEXR_KIND_DEXPILOT && kp.state && comp == 0) kp.state[r0] = my_nst;
          has = false;
        }
      }
      SPROF_STAGE(5)
    }
    SPROF_FLUSH()
    WTRACE_FLUSH()
    };
    if constexpr (TIP) {
      // Tile launches know their frames on entry: touch the lines of this lane's first frame (last_qpos, the term's two
      // keypoints) BEFORE the constants of the pass are fetched and pinned, so that the HBM round trip overlaps the table
      // set-up instead of following it (the loads proper then hit in L2): 64 frames 19.9 -> 19.5 us, 16 384 frames 25.4 ->
      // 24.9 us.  Not for full-chip launches: their initial loads are bound by the NUMBER of requests 4 096 waves put into
      // the L1s at once, and three more per lane cost more than the overlap returns (65 536 frames: 42.5 -> 43.2 us).
      float touch0 = 0.f, touch1 = 0.f, touch2 = 0.f;
      const bool touch = TIP32 && kp.qchunk == 0 && kp.kpts != nullptr && !seq && item_raw < nB &&
                         (int64_t)gridDim.x * waves_per_block < 4096;
      if (touch) {
        const int64_t r0 = row_of(item_raw);
        const int row = tb.term_ref[0];
        touch0 = kp.last[r0 * ld + tip_api0];
        touch1 = kp.kpts[(r0 * kp.n_kp + kp.h_task[row]) * 3];
        const int o = kp.h_origin[row];
        touch2 = kp.kpts[(r0 * kp.n_kp + (o >= 0 ? o : 0)) * 3];
      }
      TipTabT<real> tt;  // float32: every constant of the pass pinned in SGPRs (dexr_tip.hpp)
      tt.load(tb, tb.term_task[0], tb.term_origin[0], W + 64 * kp.lds_terms, lane);
      if (touch) asm volatile("" ::"v"(touch0), "v"(touch1), "v"(touch2));
      run(tt);
    } else if constexpr (CHAIN) {
      LocalTab<NMAX> lt;  // tables in registers for the whole kernel (see LocalTab)
      lt.load(tb);
      run(lt);
    } else {
      run(tb);
    }
    return;
  } else {
  // ---- large components: one tile of 64 frames per wave; the Hessian (up to 300 registers) is rebuilt after a
  // rejected step instead of being kept twice.
  const int n_frames = seq ? kp.T : 1;
#pragma clang loop unroll(disable)
  for (int t_seq = 0; t_seq < n_frames; ++t_seq) {
  if (t_seq > 0) nst_out = load_item(item, t_seq, nst_out);  // sequence mode: next frame, carried start point / bits
  // start point: last_qpos clipped into the box (nlopt requires lb <= x0 <= ub; seq_retarget.py:118-120 clips)
#pragma unroll
  for (int k = 0; k < NMAX; ++k)
    if ((optmask >> k) & 1u) S.x[k] = fmin(fmax(S.x[k], (real)tb.lo[k]), (real)tb.hi[k]);
  real lam = (real)kp.lam0, nu = 2;
  bool done = false;
  int status = ST_MAXITER;
  int my_iters = 0, blind = 0, nrej = 0;
  real sprev = (real)1e30;
  real xo[NMAX];
  real F;
  S.fk(tb, nj, P, lane);
  F = S.template residuals<0>(tb, kp, nt, vmask, P, T, W, lane);
#pragma unroll
  for (int k = 0; k < NMAX; ++k)
    if ((optmask >> k) & 1u) F += delta * (S.x[k] - S.xl[k]) * (S.x[k] - S.xl[k]);

  for (int it = 0; it < kp.max_iter; ++it) {
    if (__all(done)) break;
    // quadratic model at x (the FK state in ax/og/P belongs to x here)
    S.template residuals<2>(tb, kp, nt, vmask, P, T, W, lane);
    S.template fold_mimic<true>(tb, nj);
    uint32_t freemask = 0;  // lane-varying: optimised joints not held at a bound by the gradient sign
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if ((optmask >> k) & 1u) {
        S.g[k] += (real)2 * delta * (S.x[k] - S.xl[k]);
        const bool act = (S.x[k] <= (real)tb.lo[k] && S.g[k] > 0) || (S.x[k] >= (real)tb.hi[k] && S.g[k] < 0);
        if (act) S.g[k] = 0;
        else freemask |= 1u << k;
      } else {
        S.g[k] = 0;
      }
    }
    // reduced, damped system: rows/cols of held or non-variable joints become identity
    real hds = 0;
    int hdn = 0;
#pragma unroll
    for (int rr = 0; rr < NMAX; ++rr) {
      const bool fr = (freemask >> rr) & 1u;
#pragma unroll
      for (int cc = 0; cc < rr; ++cc) {
        const bool fc = (freemask >> cc) & 1u;
        if (!(fr && fc)) S.H[LS::hidx(rr, cc)] = 0;
      }
      hds += fr ? S.H[LS::hidx(rr, rr)] : (real)0;
      hdn += fr ? 1 : 0;
      S.H[LS::hidx(rr, rr)] = fr ? S.H[LS::hidx(rr, rr)] + (real)2 * delta + lam : (real)1;
    }
    const real hdmean = hds / (real)(hdn > 0 ? hdn : 1);
    real d[NMAX];
    const bool ok = S.chol_solve(d);
    // trial point (projected onto the box)
    real smax = 0, pred = 0;
    real dmax = 0, gd = 0, dd = 0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if ((optmask >> k) & 1u) {
        dmax = fmax(dmax, fabs(d[k]));
        gd -= S.g[k] * d[k];
        dd += d[k] * d[k];
      }
    // trust radius (see the small-component path): step scaled to at most step_cap per joint
    const real alpha = (kp.step_cap > 0 && dmax > (real)kp.step_cap) ? (real)kp.step_cap / dmax : (real)1;
    pred = alpha * ((real)1 - (real)0.5 * alpha) * gd + (real)0.5 * alpha * alpha * lam * dd;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      xo[k] = S.x[k];
      if ((optmask >> k) & 1u) {
        const real xt = fmin(fmax(S.x[k] + alpha * d[k], (real)tb.lo[k]), (real)tb.hi[k]);
        smax = fmax(smax, fabs(xt - S.x[k]));
        if (!done) S.x[k] = xt;
      }
    }
    // verified, undamped model and a Newton step below blind_tol: take it and stop (see the small-component path)
    const bool last_step = !done && ok && smax < (real)kp.blind_tol && lam <= (real)kp.lam0;
    S.fk(tb, nj, P, lane);
    real Ft = S.template residuals<0>(tb, kp, nt, vmask, P, T, W, lane);
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if ((optmask >> k) & 1u) Ft += delta * (S.x[k] - S.xl[k]) * (S.x[k] - S.xl[k]);

    const real noise = (real)16 * RT::eps() * fmax(fabs(F), (real)2e-3);  // see the small-component path
    const bool finite = (Ft == Ft) && (smax == smax) && (fabs(Ft) < (real)1e30);
    // below the rounding floor of F the decrease test is meaningless: trust the (small) Newton step
    const bool below_floor = ok && finite && (pred <= noise) && (smax < (real)1e-2);
    const bool accept = !done && ok && finite && ((Ft <= F) || below_floor || last_step);
    bool redo = false;
    if (!done && last_step && finite) {
      ++my_iters;
      F = Ft;
      done = true;
      status = ST_CONVERGED;
    } else if (!done) {
      ++my_iters;
      if (accept) {
        const real rho = (F - Ft) / fmax(pred, (real)1e-30);
        const real t = (real)2 * rho - (real)1;
        real shrink = below_floor ? (real)(1.0 / 3.0) : fmax((real)(1.0 / 3.0), (real)1 - t * t * t);
        if (kp.lam_fastdec > 0 && rho > (real)0.9)
          shrink = (kp.lam_recover > 0 && nrej <= 2 && lam > (real)10 * (real)kp.lam0) ? (real)kp.lam_recover : (real)kp.lam_fastdec;
        lam = fmax(lam * shrink, (real)1e-9);
        nu = 2;
        F = Ft;
        // below the floor, progress is judged by the step length alone: stop when it is under tol, when it no
        // longer contracts (rounding noise of the gradient reached), or after max_blind such steps.
        const bool stalled = below_floor && blind >= kp.stall_from && smax > (real)kp.stall_ratio * sprev && smax < (real)kp.stall_cap * (real)kp.tol;
        blind = below_floor ? blind + 1 : 0;
        sprev = smax;
        // a step below tol only means convergence when the damping is not what made it small (see dexr_big.hpp)
        const real lam_ok = fmax((real)2 * delta, (real)10 * (real)kp.lam0);
        if ((smax < (real)kp.tol && lam <= lam_ok) || stalled || blind >= kp.max_blind) {
          done = true;
          status = ST_CONVERGED;
        } else if (smax < (real)kp.tol) {
          lam = fmax((real)0.1 * lam, (real)0.5 * lam_ok);
        }
      } else {
        ++nrej;
        lam = fmax(lam, (real)1e-6) * nu;
        if (kp.lam_jump > 0) lam = fmax(lam, (real)kp.lam_jump * hdmean);
        nu *= 2;
        if (lam > (real)1e10) {  // no descent direction resolvable any more
          done = true;
          status = finite ? ST_CONVERGED : ST_FALLBACK;
        }
        if (ok && finite && smax < (real)kp.tol) {  // rejected (valid) step below tol: converged at the rounding floor of F
          done = true;
          status = ST_CONVERGED;
        }
        redo = !done;
      }
    }
    if (!accept) {
#pragma unroll
      for (int k = 0; k < NMAX; ++k) S.x[k] = xo[k];
    }
    if (__any(redo)) S.fk(tb, nj, P, lane);  // rejected lanes: bring ax/og/P back to their (unchanged) x
  }

  // non-finite guard: hand back last_qpos like the reference's RuntimeError path (optimizer.py:100-102)
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NMAX; ++k)
    if ((optmask >> k) & 1u) bad = bad || !(S.x[k] == S.x[k]);
  if (bad) status = 2;

  const int64_t r0 = row_of(item);
  const int64_t orow = seq ? (int64_t)t_seq * kp.seq_stride + r0 : r0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if ((optmask >> k) & 1u) {
      const real v = bad ? S.xl[k] : S.x[k];
      S.x[k] = v;  // (sequence mode: the value the next frame starts from)
      if (valid) {
        kp.qout[orow * ld + tb.api[k]] = (float)v;
        if (kp.qout64) kp.qout64[orow * ld + tb.api[k]] = (double)v;
      }
    }
  }
  if (valid) {
    if (kp.kind == DEXR_KIND_DEXPILOT && kp.state && comp == 0 && t_seq + 1 == n_frames) kp.state[r0] = nst_out;
    if (kp.status) atomicMax(&kp.status[orow], status);
    if (kp.iters) atomicMax(&kp.iters[orow], my_iters);
    if (kp.fval) atomicAdd(&kp.fval[orow], (float)F);
  }
  }  // frames of the sequence
  }  // large components
}

}  // namespace dexr
