// dexr_wide.hpp -- solve kernel for LARGE DENSE components (9..32 joints): SIXTEEN LANES PER FRAME.
//
// Why: a launch of the four-lanes-per-frame kernel (dexr_quad.hpp) over 65 536 Shadow-DexPilot frames is bound by its
// SLOWEST frame, not by its throughput -- a frame that needs 44 iterations is a serial chain of ~56 wave passes of
// ~37 000 instructions each (0.12 ms per pass, 6.5 ms per launch, while the mean frame needs 5 iterations; see
// DESIGN.md section 4).  A pass therefore has to become short.  Here a frame is spread over a 16-lane DPP row, a wave
// holds four frames, and every stage of a pass is parallel over the 16 lanes:
//   * sines / cosines of all joints (float64), one or two joints per lane; then forward kinematics (float64): lane l
//     walks root-to-leaf chain l of the kinematic tree (one finger each; the shared wrist / free-base prefix is
//     recomputed by every lane), 5-13 joints deep instead of all 24-30 in sequence;
//   * residuals, Huber weights, DexPilot targets: lane t evaluates term t (<= 16 terms per component);
//   * Jacobian: lane l forms the columns of joints l and l + 16 for the term in flight, accumulates their gradient
//     entries and second-order vectors, and publishes the four rows (three coordinates + the Huber rank-one row,
//     pre-multiplied by the square roots of their weights) of the term's Jacobian in LDS;
//   * Hessian: 2-D cyclic over a 4 x 4 lane grid -- lane (a, b) owns H[r][c], r = a (mod 4), c = b (mod 4), 21 floats
//     at n = 24, kept as register pairs (v_pk_fma_f32) -- accumulated as outer products of the published rows; the
//     second-order (Newton) term is a_c . CF_r for every revolute ancestor c of r, added once per pass from per-joint
//     sums CF_r;
//   * Cholesky on that grid: the pivot travels by a DPP quad broadcast + a stride-4 DPP sum, the pivot column by one DPP
//     quad broadcast (row side) and one ds_bpermute (column side) per local row; the right-hand side rides along as an
//     extra matrix row (forward substitution for free); backward substitution with stride-4 DPP sums.
// The accepted point's Hessian stays in registers (its gradient in LDS), so a REJECTED step costs no extra pass (the
// quad kernel re-assembles): every pass evaluates a new trial point.
// MIMIC instantiation: the grid is that of the <= 16 optimised VARIABLES; lane v forms and sums the columns of the <= 3
// joints that move with variable v, the second-order term comes from a host-built list of (joint, revolute ancestor)
// pairs bucketed by the lane that owns the target entry.  MODCHOL (with MIMIC): the damping rules of dexr_red.hpp.
// Budgets: 256 VGPRs and 15-20 KB of LDS per wave = two waves per SIMD; the 16-row joint grid is built for three (168
// VGPRs, 13 KB).  Everything that is read once or twice per pass lives in LDS or the tables, not in registers.
// Damping / termination rules are those of dexr_quad.hpp.
// SPRINT instantiations (round 5): one frame per wave for small batches and short sequences -- the four rows share the frame's
// term loop and each tries its own damping value per pass (see the comment at the kernel).
#pragma once

#include "dexr_big.hpp"  // sincos_f64

namespace dexr {

template <int CTRL>
static __device__ __forceinline__ float wdpp(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
static __device__ __forceinline__ double wdpp64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// all-reduce over a 16-lane row; every stage adds partners symmetrically, so the 16 lanes end with identical bits
static __device__ __forceinline__ float row_sum(float v) {
  v += wdpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += wdpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += wdpp<0x141>(v);  // row_half_mirror: quad 0 <-> 1, 2 <-> 3
  v += wdpp<0x140>(v);  // row_mirror: half 0 <-> 1
  return v;
}
static __device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, wdpp<0xB1>(v));
  v = fmaxf(v, wdpp<0x4E>(v));
  v = fmaxf(v, wdpp<0x141>(v));
  v = fmaxf(v, wdpp<0x140>(v));
  return v;
}
static __device__ __forceinline__ double row_sum64(double v) {
  v += wdpp64<0xB1>(v);
  v += wdpp64<0x4E>(v);
  v += wdpp64<0x141>(v);
  v += wdpp64<0x140>(v);
  return v;
}
static __device__ __forceinline__ float wquad_sum(float v) {
  v += wdpp<0xB1>(v);
  v += wdpp<0x4E>(v);
  return v;
}
// sum over the four lanes {b, b+4, b+8, b+12} of a row (same column class)
static __device__ __forceinline__ float stride4_sum(float v) {
  v += wdpp<0x128>(v);  // row_ror:8
  v += wdpp<0x124>(v);  // row_ror:4
  return v;
}
template <int Q>
static __device__ __forceinline__ float wquad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xF, 0xF, true));
}

// -DDEXR_WIDE_PROF=1 (tools/prof_wide_stages.sh; never in the shipped library): wave 0 of block 0 accumulates the cycles
// (s_memtime) of every stage of its passes and adds them to kp.g64out[stage] when it retires.
// -DDEXR_WIDE_DIAG=1 (tools/pass_composition.sh; never in the shipped library): the per-frame iteration count written to
// kp.iters carries, in its upper bytes, how many of the frame's passes were rejected steps, steps cut by the trust radius
// and failed factorisations.
#ifdef DEXR_WIDE_DIAG
#define WDIAG(x) x
#else
#define WDIAG(x)
#endif
#ifdef DEXR_WIDE_PROF
#define WPROF_DECL long long wp_t0 = 0, wp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WPROF_START() wp_t0 = clock64()
#define WPROF_STAGE(i)                    \
  {                                       \
    const long long wp_t1 = clock64();    \
    wp_acc[i] += wp_t1 - wp_t0;           \
    wp_t0 = wp_t1;                        \
  }
#else
#define WPROF_DECL
#define WPROF_START()
#define WPROF_STAGE(i)
#endif

typedef float wv2 __attribute__((ext_vector_type(2)));  // register pair: v_pk_fma_f32 / v_pk_mul_f32 operands

#ifndef DEXR_WIDE_MINW
#define DEXR_WIDE_MINW 2
#endif

// blockDim.x = 256 (4 waves x 4 frames); dynamic LDS per wave = wide_lds_bytes(NMAX)
// NMAX: size of the Hessian grid (joints; with MIMIC: optimised VARIABLES, mimic joints folded into their source's
// column).  NJ: joints the kinematics can hold.
template <int NMAX, bool MIMIC>
struct WideLds {
  static constexpr int NJ = MIMIC ? 32 : NMAX;
  static constexpr int NR = NMAX / 4;
  static constexpr int NRP = NR <= 4 ? 4 : 8;          // class chunk of a Jacobian row, padded for 16-byte reads
  static constexpr int XT = 0;                         // NJ x 16 floats: X[12], fbeg | fend << 8, var + 1, vmul, off
  // (row stride of XT in words: 20, not 16 -- rows 256 B apart share their banks, and the kinematic chains' lanes read the rows of
  // four to five different joints with one ds_read_b128; the 16-row grids' LDS budget is exact, they keep 16)
  static constexpr int XTS = (!MIMIC && NMAX > 16) ? 20 : 16;
  static constexpr int FO = XT + NJ * XTS * 4;         // 16 frames x 4 floats
  static constexpr int CH = FO + 256;                  // 16 x 16 chain bytes
  static constexpr int ANC = CH + 256;                 // NMAX words: revolute ancestors-or-self of each joint
  static constexpr int BOX = ANC + NMAX * 4;           // NMAX x (lo, hi): box of each grid variable
  static constexpr int PL = BOX + NMAX * 8;            // MIMIC: second-order pair list, 128 words + 17 lane offsets
  static constexpr int TM = PL + (MIMIC ? 512 + 32 : 0);  // MIMIC: 16 terms x (ancestor mask of the task frame, of the
                                                       // origin frame); the joint-space grids keep them in the spare
                                                       // words 13 / 14 of XT row t
  static constexpr int SLOT0 = TM + (MIMIC ? 128 : 0);
  // per frame slot
  static constexpr int P = 0;                          // 16 frames x 3 doubles
  static constexpr int AX = P + 384;                   // NJ x 4 floats
  static constexpr int OG = AX + NJ * 16;              // NJ x 4 floats
  static constexpr int SC = OG + NJ * 16;              // NJ x 2 doubles: (sin q, cos q) of a revolute joint, (q, -) of a
                                                       // prismatic one, computed by the joint's slot lane
  static constexpr int XV = SC + NJ * 16;              // NJ floats: joint values of the trial point (MIMIC: variables)
  static constexpr int QJ = XV + NJ * 4;               // MIMIC: NJ floats, values of the fixed joints of the frame
  static constexpr int GV = QJ + (MIMIC ? NJ * 4 : 0); // NMAX floats: gradient
  static constexpr int CF = GV + NMAX * 4;             // NJ x 4 floats: second-order vectors
  static constexpr int TB = CF + NJ * 16;              // 16 terms x 16 floats (MIMIC: reused for the second-order sums)
  static constexpr int JR = TB + 1024;                 // 3 rows x 4 classes x NRP floats
  static constexpr int FS = JR + 3 * 4 * NRP * 4;      // 32 bytes: row of the frame's inputs / of its item, item, frame of
                                                       // the sequence, DexPilot bits (registers are the scarce resource)
  static constexpr int XL = FS + 32;                   // NMAX floats: regularisation target (the frame's start row)
  // (TGLDS: the terms' target vectors / weights in 256 B of the frame slot instead of four registers of the term's lane.  It
  // takes the 24-row grid from 2 spilled VGPRs to none -- and 1 KB of LDS per wave, 8 KB per CU, which is what the small-component
  // kernels of a FLEET batch need to sit beside the heavy model's blocks: same box, fleet step 0.750 -> 0.79-0.85 ms
  // (profiles/r05_fleet_ab_tg_lds_same_box.txt).  Off.)
  static constexpr bool TGLDS = false;
  static constexpr int TG = XL + NMAX * 4;
  static constexpr int SLOT = TG + (TGLDS ? 256 : 0);
  // LDS decides the occupancy: two blocks of four waves per CU (160 KB); the 16-row joint grid is built for three
  static_assert(2 * 4 * (SLOT0 + 4 * SLOT) <= 160 * 1024, "two blocks per CU must fit");
  static_assert(MIMIC || NMAX != 16 || 3 * (4 * (SLOT0 + 4 * SLOT) + 512) <= 160 * 1024, "three blocks per CU must fit");
  static constexpr int WAVE = SLOT0 + 4 * SLOT;
};

// SPRINT (round 5; every grid): ONE FRAME PER WAVE -- the launch shape of small batches (the reference's one-frame-per-
// call loop above all), where a wave's four rows would otherwise hold one frame and three idle copies of the instruction
// stream.  All four rows hold the SAME frame and run kinematics, terms, factorisation and the step logic redundantly (identical
// instructions on identical inputs: identical bits, no exchange needed), but the TERM LOOP -- 40 % of a DexPilot pass -- is split:
// row s forms the columns and outer products of terms s, s + 4, ... and the partial Hessians / gradients / second-order vectors
// are summed across the rows by an xor butterfly (ds_bpermute, lane ^ 16, lane ^ 32: every row ends with the same bits).  Same
// damping rules, same sequence of trial points up to the summation order of the Hessian; only row 0 writes results.
template <int NMAX, bool MIMIC, bool MODCHOL, bool SPRINT = false>
__global__ void __launch_bounds__(256, DEXR_WIDE_MINW) dexr_wide_kernel(const KernelParams kp, const dexr_comp_table* __restrict__ comps,
                                                                         const WideTable* __restrict__ wtabs) {
  static_assert(NMAX == 16 || NMAX == 24 || NMAX == 32, "bucket");
  static_assert(!MIMIC || NMAX == 16, "the variable grid of the mimic kernel has 16 rows");
  constexpr int FPW = SPRINT ? 1 : 4;  // frames a wave holds at a time
  using L = WideLds<NMAX, MIMIC>;
  constexpr int NJ = L::NJ;
  constexpr int NR = NMAX / 4;
  constexpr int NRP = L::NRP;
  constexpr int NJ2 = NMAX > 16 ? 2 : 1;  // grid indices owned per lane (l and l + 16): joints, with MIMIC variables
  constexpr int NFS = 2;                  // MIMIC: joint slots per lane (l, l + 16) for the fixed joints' values
  constexpr int FAM = 3;                  // MIMIC: joints per variable (the variable's own joint + its mimic followers)
  // Optional damping dynamics of dexr_red.hpp: a non-positive pivot is reflected instead of failing the pass, such a
  // step is judged by the decrease alone and stretched towards the trust radius, and a rejection raises lambda to
  // lam_jump x mean diag(H).  Measured (65 536 frames): shorter iteration tails for DexPilot models with mimic joints
  // (Inspire 1.35-1.48 -> 0.86-0.99 ms), longer ones for position models (Inspire 1.24-1.36 -> 1.66-1.78 ms) and for the
  // joint-space models (Shadow DexPilot, round 2) -- hence a choice per model (dexr_tuning.pivot_rule; instantiated for the
  // variable-grid kernel only: as a run-time flag it cost the joint-space kernels 56 more spilled registers).

  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  int l = lane & 15;        // lane within the frame's row
  const int slot = lane >> 4;     // frame slot of the wave
  int a = l >> 2, b = l & 3;  // row / column class of the Hessian grid
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * waves_per_block + wave_in_block;
  const int comp = (int)(wave_global % kp.n_comp);
  const int64_t tile = wave_global / kp.n_comp;
  const int rowbase4 = (lane & 48) << 2;  // ds_bpermute byte address of lane 0 of this row

  unsigned char* wbase = lds_raw + (size_t)wave_in_block * L::WAVE;
  float* XT = reinterpret_cast<float*>(wbase + L::XT);
  float* FO = reinterpret_cast<float*>(wbase + L::FO);
  unsigned char* CH = wbase + L::CH;
  uint32_t* ANCw = reinterpret_cast<uint32_t*>(wbase + L::ANC);
  float* BOXw = reinterpret_cast<float*>(wbase + L::BOX);
  uint32_t* PLw = reinterpret_cast<uint32_t*>(wbase + L::PL);
  unsigned char* POFF = wbase + L::PL + 512;
  // per-term ancestor masks (mt, mo): read with the term's block at the top of every iteration of the term loop instead of
  // two DEPENDENT scalar loads from the tables (term_task -> frame_anc: ~2 scalar-memory round trips per term and pass)
  uint32_t* TMw = MIMIC ? reinterpret_cast<uint32_t*>(wbase + L::TM) : reinterpret_cast<uint32_t*>(wbase + L::XT) + 13;
  constexpr int XTS = L::XTS;
  constexpr int TMS = MIMIC ? 2 : XTS;  // words between consecutive terms' mask pairs
  unsigned char* sbase = wbase + L::SLOT0 + (size_t)slot * L::SLOT;
  double* Pl = reinterpret_cast<double*>(sbase + L::P);
  float* AXl = reinterpret_cast<float*>(sbase + L::AX);
  float* OGl = reinterpret_cast<float*>(sbase + L::OG);
  double* SCl = reinterpret_cast<double*>(sbase + L::SC);
  float* XVl = reinterpret_cast<float*>(sbase + L::XV);
  float* QJl = reinterpret_cast<float*>(sbase + L::QJ);
  float* GVl = reinterpret_cast<float*>(sbase + L::GV);
  float* CFl = reinterpret_cast<float*>(sbase + L::CF);
  float* TBl = reinterpret_cast<float*>(sbase + L::TB);
  float* JRl = reinterpret_cast<float*>(sbase + L::JR);
  int64_t* FS64 = reinterpret_cast<int64_t*>(sbase + L::FS);     // [0] irow, [1] lrow, [2] item
  int32_t* FS32 = reinterpret_cast<int32_t*>(sbase + L::FS) + 6;  // [0] frame of the sequence, [1] DexPilot bits
  float* XLl = reinterpret_cast<float*>(sbase + L::XL);
  float* TGl = reinterpret_cast<float*>(sbase + L::TG);

  const dexr_comp_table& tb = comps[comp];
  const WideTable& wt = wtabs[comp];
  const int nj = tb.n_joint, nt = tb.n_term;
  const int ng = MIMIC ? tb.n_var : nj;  // rows of the Hessian grid in use
  const int depth = wt.depth;
  const bool fk_rows3 = wt.n_chain <= 5;  // three lanes per kinematic chain (see fk)
  const float delta = kp.norm_delta;
  const int64_t nB = kp.bucket ? (int64_t)kp.bucket[1] : kp.B;
  const int64_t pbase = kp.bucket ? (int64_t)kp.bucket[0] : 0;
  auto row_of = [&](int64_t it) -> int64_t { return kp.perm ? (int64_t)kp.perm[pbase + it] : it; };
  const int ld = kp.ld;
  const bool seq = kp.T > 0;
  WPROF_DECL
  // COLD PATHS READ THE KERNEL ARGUMENTS AGAIN.  Taking a frame and retiring it happen once per frame, a pass 5-40 times; the
  // ~25 kernel arguments only those two need (batch pointers, strides, DexPilot distances, the keypoint map) were loaded once,
  // kept in SGPRs for the whole pass loop and, there being ~100 of those, spilled to VGPR lanes (259 spilled SGPRs = 5 VGPRs of a
  // kernel at 256, `v_readlane`s all over the loop).  Through a pointer the compiler cannot see through (the kernarg segment,
  // kp is the first argument; address space 4 keeps the reads scalar loads) they are s_loads inside the cold path instead.
  typedef const __attribute__((address_space(4))) KernelParams ColdParams;
  auto cold = [&]() -> ColdParams& {
    ColdParams* q = (ColdParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return *q;
  };

  // ---- wave-constant tables into LDS (lane-varying joint indices read them in the kinematics) ----------------------
  for (int k = lane; k < NJ; k += 64) {
    const bool in = k < nj;
#pragma unroll
    for (int i = 0; i < 12; ++i) XT[k * XTS + i] = in ? tb.X[k][i] : 0.f;
    XT[k * XTS + 12] = __int_as_float(in ? (tb.fbeg[k] | (tb.fend[k] << 8)) : 0);
    if (MIMIC) {
      XT[k * XTS + 13] = __int_as_float(in ? tb.var[k] + 1 : 0);  // 0: not driven by a variable (fixed joint)
      XT[k * XTS + 14] = in ? tb.vmul[k] : 0.f;
      XT[k * XTS + 15] = in ? tb.off[k] : 0.f;
    } else {
      XT[k * XTS + 15] = 0.f;
    }
  }
  for (int t = lane; t < 16; t += 64) {
    const bool in = t < nt;
    const int ft = in ? tb.term_task[t] : 0, fo = in ? tb.term_origin[t] : -1;
    TMw[t * TMS] = in ? tb.frame_anc[ft] : 0u;
    TMw[t * TMS + 1] = (in && fo >= 0) ? tb.frame_anc[fo] : 0u;
  }
  if (MIMIC) {
    for (int i = lane; i < 128; i += 64) PLw[i] = wt.pair[i];
    for (int i = lane; i < 32; i += 64) POFF[i] = i < 17 ? wt.pair_off[i] : 0;
  }
  for (int f = lane; f < 16; f += 64) {
    const bool in = f < tb.n_frame;
#pragma unroll
    for (int i = 0; i < 3; ++i) FO[f * 4 + i] = in ? tb.frame_off[f][i] : 0.f;
    // spare word of row f: the keypoint map of REFERENCE ROW f (task | origin << 8, 0xFF: no origin) -- read when a frame is
    // taken (ref_row), from LDS instead of through a dependent global load in front of every keypoint read
    uint32_t hm = 0xFF00u;
    if (kp.kpts && f < kp.n_ref) {
      const int o = kp.h_origin[f];
      hm = (uint32_t)(kp.h_task[f] & 0xFF) | ((uint32_t)(o >= 0 ? o : 0xFF) << 8);
    }
    FO[f * 4 + 3] = __int_as_float((int)hm);
  }
  for (int i = lane; i < 256; i += 64) CH[i] = wt.chain[i >> 4][i & 15];
  // (sincos_f64_tab's 512-byte table is read from global memory through the vector L1: a copy in the wave's LDS measured the
  // same, profiles/r06_wide_ab_sincos_table_same_box.txt)
  for (int i = lane; i < NMAX; i += 64) ANCw[i] = (!MIMIC && i < nj) ? wt.anc_rev[i] : 0u;

  uint32_t revmask = 0;
#pragma unroll
  for (int k = 0; k < NJ; ++k)
    if (k < nj && tb.jtype[k] == DEXR_JOINT_REVOLUTE) revmask |= 1u << k;

  const bool any_prismatic = revmask != (nj >= 32 ? 0xFFFFFFFFu : ((1u << nj) - 1u));  // wave-uniform
  // ---- per-lane constants: the joints this lane owns (l, l + 16) and the ancestor masks of its Hessian rows ---------
  int jo_[NJ2];
  bool jin[NJ2], jopt[NJ2], jrev[NJ2], jfix[NJ2];
  // (the joint whose box / api index / fixed-value map apply is looked up in the tables when a frame is loaded or
  // retired: registers are the scarce resource of this kernel)
  auto jsel = [&](int s) -> int { return jin[s] ? (MIMIC ? tb.var_joint[jo_[s]] : jo_[s]) : 0; };
#pragma unroll
  for (int s = 0; s < NJ2; ++s) {
    const int k = l + 16 * s;  // grid index: joint, with MIMIC variable
    jo_[s] = k;
    jin[s] = k < ng;
    const int kk = jin[s] ? (MIMIC ? tb.var_joint[k] : k) : 0;  // the joint whose box / api index apply
    jopt[s] = jin[s] && tb.src_kind[kk] == DEXR_SRC_OPT;
    jfix[s] = !MIMIC && jin[s] && tb.src_kind[kk] == DEXR_SRC_FIXED;
    jrev[s] = jin[s] && tb.jtype[kk] == DEXR_JOINT_REVOLUTE;
    if (jin[s]) {  // (every row's lane l writes the same two values: benign)
      BOXw[2 * k] = tb.lo[kk];
      BOXw[2 * k + 1] = tb.hi[kk];
    }
  }
  // MIMIC: the joints that move with this lane's variable (its own joint first) and their dq/dx; the fixed joints in
  // this lane's joint slots (their values go to LDS once per frame)
  int famk[FAM];
  float famm[FAM];
  bool fsfix[NFS];

  int fam_max = 0;  // wave-uniform: the largest family of the component
  if (MIMIC) {
#pragma unroll
    for (int e = 0; e < FAM; ++e) {
      const unsigned kb = wt.fam[l][e];
      famk[e] = kb == 0xFFu ? -1 : (int)kb;
      famm[e] = kb == 0xFFu ? 0.f : tb.vmul[kb];
    }
    fam_max = wt.fam_max;
#pragma unroll
    for (int s = 0; s < NFS; ++s) {
      const int k = l + 16 * s;
      fsfix[s] = k < nj && tb.src_kind[k < nj ? k : 0] == DEXR_SRC_FIXED;
    }
  }

  const bool per_coord = kp.kind == DEXR_KIND_POSITION;
  const double beta = (double)kp.huber_delta, ibeta = 1.0 / beta;
  const bool newton = kp.newton != 0;
  const bool dexpilot = kp.kind == DEXR_KIND_DEXPILOT;
  const int F_ = kp.num_fingers, n_pair = F_ * (F_ - 1) / 2, len_s1 = F_ - 1;

  // ---- per-frame state (replicated in the 16 lanes of the row unless noted) ------------------------------------------
  // per-frame bookkeeping lives in LDS (FS64 / FS32), read where it is needed
  auto f_irow = [&]() -> int64_t { return FS64[0]; };
  auto f_lrow = [&]() -> int64_t { return FS64[1]; };
  auto f_nst = [&]() -> uint32_t { return (uint32_t)FS32[1]; };
  bool active = false;
  float xj[NJ2], xacc[NJ2];  // own joints: trial value, accepted value (the regularisation target is in LDS: XLl)
  // (the gradient at the accepted point, incl. regulariser, lives in LDS: GVl)
  constexpr int NP = NR / 2;             // column pairs of the local Hessian block
  wv2 Ha[NR][NP];                        // Hessian grid entries (4 i + a, 4 j + b), j <= i, at the accepted point;
                                         // pair jj holds local columns 2 jj, 2 jj + 1 (packed FMAs)
#pragma unroll
  for (int s = 0; s < NJ2; ++s) { xj[s] = 0; xacc[s] = 0; }
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NP; ++j) Ha[i][j] = wv2{0.f, 0.f};

  // KPLDS (round 6): a row that takes a frame copies the frame's input block -- the 21 raw keypoints (252 B) or its n_ref
  // ready-made rows -- into the (then idle) term-block area of its slot with ONE coalesced round of loads, next to the loads of
  // last_qpos and the DexPilot bits; the projection test and the targets then read LDS.  Before, the hand-out was a chain of
  // dependent global round trips (keypoint map -> keypoints for the projection bits -> map -> keypoints for the targets):
  // 5-7 k of a pass's ~43 k cycles at 65 536 frames (tools/prof_wide_stages.sh), and the 12-byte reads fetched every line of
  // the block two or three times.  (Not at n = 32, whose terms re-read their targets in every pass -- TGREG.)
  constexpr bool KPLDS = NMAX <= 24;
  auto ref_row = [&](ColdParams& kp, int row, float (&rv)[3]) {
    if (KPLDS) {
      const float* blk = TBl;
      if (kp.kpts) {
        const uint32_t hm = (uint32_t)__float_as_int(FO[row * 4 + 3]);
        const int ta = (int)(hm & 0xFFu), o = (int)((hm >> 8) & 0xFFu);
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = blk[ta * 3 + i] - (o != 0xFF ? blk[(o != 0xFF ? o : 0) * 3 + i] : 0.f);
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = blk[row * 3 + i];
      }
      return;
    }
    const int64_t irow = f_irow();
    if (kp.kpts) {
      // (32-bit element offsets inside the frame's keypoint block: as 64-bit byte offsets of a lane's own rows they are
      // loop invariants the compiler keeps in VGPR pairs across the pass loop -- two of them went to scratch)
      const float* frame = kp.kpts + irow * (int64_t)(kp.n_kp * 3);
      const float* pa = frame + kp.h_task[row] * 3;
      const int o = kp.h_origin[row];
      if (o >= 0) {
        const float* pb = frame + o * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = pa[i] - pb[i];
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = pa[i];
      }
    } else {
      const float* r = kp.ref + (irow * kp.n_ref + row) * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) rv[i] = r[i];
    }
  };
  auto xl = [&](ColdParams& kp, int s, const float* lastp, int t_seq) -> float {  // regularisation target of own joint s (see dexr_quad.hpp)
    float v;
    if (seq && t_seq > 0)
      v = __hip_atomic_load(const_cast<float*>(lastp) + tb.api[jsel(s)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      v = lastp[tb.api[jsel(s)]];
    return seq ? fminf(fmaxf(v, BOXw[2 * jo_[s]] + kp.clip_eps), BOXw[2 * jo_[s] + 1] - kp.clip_eps) : v;
  };
  // Target vector and weight of the lane's own term (lane t evaluates term t in every pass): constant while the row's frame
  // is being solved, so they are formed ONCE when the frame is taken and kept in four registers of that lane -- re-reading
  // the keypoints in every pass put a global-load round trip into each pass and was the 1.5 x between the kernel's HBM
  // traffic and its algorithmic bytes.  (Not at n = 32, where the Hessian grid already spills.)
  constexpr bool TGREG = NMAX <= 24;
  constexpr bool TGLDS = L::TGLDS;  // ... in 16 bytes of the frame slot per term where the grid leaves no registers (n = 24)
  float tgt[4] = {0.f, 0.f, 0.f, 1.f};
  // frames of the lane's own term (table reads indexed by the lane: taken once, here, where the lane index is still a
  // loop-invariant the compiler may use -- inside the pass loop it is opaque, see the top of the loop)
  const int my_ft = l < nt ? tb.term_task[l] : 0, my_fo = l < nt ? tb.term_origin[l] : -1;
  const int my_ref = l < nt ? tb.term_ref[l] : 0;
  auto load_frame = [&](int64_t it, int t) {
    ColdParams& kp = cold();  // (shadows the kernel argument inside this cold path)
    const int t_seq = t;
    const int64_t lrow = row_of(it);
    const int64_t irow = seq ? (int64_t)t * kp.seq_stride + lrow : lrow;
    const float* lastp = (seq && t > 0) ? kp.qout + (irow - kp.seq_stride) * ld : kp.last + lrow * ld;
    uint32_t nst = (seq && t > 0) ? f_nst() : 0u;  // (sequence mode: the previous frame's bits are the carried state)
    FS64[0] = irow;
    FS64[1] = lrow;
    FS64[2] = it;
    FS32[0] = t;
    // every global read of the hand-out is issued here, in one round: the input block, last_qpos, the state word
    float xl_pre[NJ2];
#pragma unroll
    for (int s = 0; s < NJ2; ++s) xl_pre[s] = jopt[s] ? xl(kp, s, lastp, t_seq) : 0.f;
    const uint32_t st_pre = (dexpilot && !(seq && t_seq > 0) && kp.state) ? kp.state[lrow] : 0u;
    if (KPLDS) {
      const float* src = kp.kpts ? kp.kpts + irow * (int64_t)(kp.n_kp * 3) : kp.ref + irow * (int64_t)(kp.n_ref * 3);
      const int cnt = kp.kpts ? kp.n_kp * 3 : kp.n_ref * 3;
      float blk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) blk[i] = (l + 16 * i < cnt) ? src[l + 16 * i] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (l + 16 * i < cnt) TBl[l + 16 * i] = blk[i];
      for (int i = l + 64; i < cnt; i += 16) TBl[i] = src[i];  // (more than 21 keypoints per frame)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < NJ2; ++s) {
      xj[s] = 0;
      float xlast = 0;
      if (jopt[s]) {
        xlast = xl_pre[s];
        const float v = (kp.x0 && !(seq && t_seq > 0)) ? kp.x0[lrow * ld + tb.api[jsel(s)]] : xlast;
        xj[s] = fminf(fmaxf(v, BOXw[2 * jo_[s]]), BOXw[2 * jo_[s] + 1]);
      } else if (jfix[s]) {
        xj[s] = tb.mult[jsel(s)] * kp.fixed[irow * kp.ldf + tb.src_idx[jsel(s)]] + tb.off[jsel(s)];
      }
      xacc[s] = xj[s];
      if (jin[s]) XLl[jo_[s]] = xlast;
    }
    if (MIMIC) {
#pragma unroll
      for (int s = 0; s < NFS; ++s)
        if (fsfix[s]) {
          const int k = l + 16 * s;
          QJl[k] = tb.mult[k] * kp.fixed[irow * kp.ldf + tb.src_idx[k]] + tb.off[k];
        }
    }
    if (dexpilot) {  // projection bits (optimizer.py:466-476)
      // One lane per pair vector (n_pair <= 10 of the row's 16 lanes): its length from ONE round of keypoint loads, the
      // thumb-finger bits (S1, hysteresis on the incoming state) gathered with a row ballot, then the finger-finger bits
      // (S2 = both S1 bits and a short vector).  A loop over the rows in every lane, as the other kernels have it, waits
      // for ten dependent load round trips each time a row takes a new frame: tools/prof_wide_stages.sh showed 5 k of a
      // pass's 39 k cycles going to this hand-out at 65 536 frames.
      const uint32_t st = (seq && t_seq > 0) ? nst : st_pre;
      float dist = 0.f;
      if (l < n_pair) {
        float rv[3];
        ref_row(kp, l, rv);
        dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
      }
      bool bb = false;
      if (l < len_s1) {
        bb = (st >> l) & 1u;
        if (dist < kp.project_dist) bb = true;
        if (dist > kp.escape_dist) bb = false;
      }
      const uint32_t s1 = (uint32_t)(__ballot(bb) >> (16 * slot)) & 0xFFFFu;
      bool b2b = false;
      if (l >= len_s1 && l < n_pair) {
        int aa = 0, b2 = 1;  // pair l - len_s1 of the enumeration aa < b2 < F - 1, aa-major (optimizer.py:442-447)
        for (int i = len_s1; i < l; ++i) {
          ++b2;
          if (b2 >= F_ - 1) {
            ++aa;
            b2 = aa + 1;
          }
        }
        b2b = ((s1 >> b2) & 1u) && ((s1 >> aa) & 1u) && (dist <= 0.03f);
      }
      nst = s1 | ((uint32_t)(__ballot(b2b) >> (16 * slot)) & 0xFFFFu);
    }
    FS32[1] = (int32_t)nst;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // target vector and weight of one term (optimizer.py:246, 479-507)
  auto term_target = [&](ColdParams& kp, int row, float (&tv)[3], float& wgt) {
    float rv[3];
    ref_row(kp, row, rv);
    wgt = 1.f;
    if (dexpilot) {
      if (row < n_pair) {
        if ((f_nst() >> row) & 1u) {
          const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
          const float eta = row < len_s1 ? kp.eta1 : kp.eta2;
#pragma unroll
          for (int i = 0; i < 3; ++i) tv[i] = (rv[i] / (dist + 1e-6f)) * eta;
          wgt = row < len_s1 ? 200.f : 400.f;
          return;
        }
      } else {
        wgt = (float)(n_pair + F_);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) tv[i] = rv[i] * kp.scaling;
    } else {
      const float sc = (kp.kind == DEXR_KIND_VECTOR) ? kp.scaling : 1.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) tv[i] = rv[i] * sc;
    }
  };

  // (called by every lane of a row right after load_frame)
  auto load_target = [&]() {
    if (TGREG && l < nt) {
      float tv[3], wgt;
      int row = my_ref;
      asm volatile("" : "+v"(row));  // (opaque: the 64-bit offsets of this row into the keypoint map are formed here, in the
                                     // cold path, not kept -- in scratch -- across the pass loop)
      term_target(cold(), row, tv, wgt);
      if (TGLDS) {
        *reinterpret_cast<float4*>(TGl + l * 4) = make_float4(tv[0], tv[1], tv[2], wgt);  // (read by the same lane only)
      } else {
        tgt[0] = tv[0]; tgt[1] = tv[1]; tgt[2] = tv[2]; tgt[3] = wgt;
      }
    }
  };

  // idle rows run the passes on stale data: their bookkeeping block must address valid rows (row 0) from the start
  FS64[0] = 0;
  FS64[1] = 0;
  FS64[2] = 0;
  FS32[0] = 0;
  FS32[1] = 0;
  if (l < NMAX) XLl[l] = 0.f;
  if (NJ2 > 1 && l + 16 < NMAX) XLl[l + 16] = 0.f;
  if (TGLDS) *reinterpret_cast<float4*>(TGl + l * 4) = make_float4(0.f, 0.f, 0.f, 1.f);
  // frames on the fixed base never move
  if (l < tb.n_base_frame) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Pl[l * 3 + i] = (double)tb.frame_off[l][i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- float64 forward kinematics, one root-to-leaf chain per lane --------------------------------------------------
  auto fk = [&]() {
#pragma unroll
    for (int s = 0; s < NJ2; ++s)
      if (jin[s]) XVl[jo_[s]] = xj[s];
    if (MIMIC) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // sines / cosines of all joints first, one or two joints per lane: the chain walk below then has no polynomial in
    // its dependency chain (a chain is 7-13 joints deep; every lane used to evaluate all of its chain's sincos in turn)
#pragma unroll
    for (int s = 0; s < (NJ + 15) / 16; ++s) {
      const int k = l + 16 * s;
      if (k < nj) {
        double q;
        if (MIMIC) {  // q = vmul * x[var] + off in float64 (a float32 product makes F a step function of x)
          const float4 x3 = *reinterpret_cast<const float4*>(XT + k * XTS + 12);
          const int vc = __float_as_int(x3.y);
          q = vc > 0 ? fma((double)x3.z, (double)XVl[vc > 0 ? vc - 1 : 0], (double)x3.w) : (double)QJl[k];
        } else {
          q = (double)xj[s < NJ2 ? s : 0];
        }
        double sn = q, cs = 0.0;
        if ((revmask >> k) & 1u) sincos_f64_tab(q, &sn, &cs);  // (table + short polynomials: dexr_math.hpp)
        SCl[2 * k] = sn;
        SCl[2 * k + 1] = cs;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (fk_rows3) {
      // <= 5 chains (every shipped hand): THREE lanes per chain, lane 3 c + i carries row i of the chain's rotation and
      // coordinate i of its position -- a joint costs 16 float64 operations per lane instead of 48, and 15 of the 16
      // lanes work instead of 5
      const int ch = l / 3, ri = l - 3 * ch;
      double r0 = ri == 0 ? 1.0 : 0.0, r1 = ri == 1 ? 1.0 : 0.0, r2 = ri == 2 ? 1.0 : 0.0, pi = 0.0;
      unsigned cb = depth > 0 ? CH[ch * 16] : 0xFFu;
      int kf = cb != 0xFFu ? (int)(cb & 0x7Fu) : 0;
      float4 x0 = *reinterpret_cast<const float4*>(XT + kf * XTS);
      float4 x1 = *reinterpret_cast<const float4*>(XT + kf * XTS + 4);
      float4 x2 = *reinterpret_cast<const float4*>(XT + kf * XTS + 8);
      float4 x3 = *reinterpret_cast<const float4*>(XT + kf * XTS + 12);
      double scs = SCl[2 * kf], scc = SCl[2 * kf + 1];
#pragma clang loop unroll(disable) vectorize(disable)
      for (int s = 0; s < depth; ++s) {
        const unsigned cbn = (s + 1 < depth) ? CH[ch * 16 + s + 1] : 0xFFu;
        const int kn = cbn != 0xFFu ? (int)(cbn & 0x7Fu) : 0;
        const float4 y0 = *reinterpret_cast<const float4*>(XT + kn * XTS);
        const float4 y1 = *reinterpret_cast<const float4*>(XT + kn * XTS + 4);
        const float4 y2 = *reinterpret_cast<const float4*>(XT + kn * XTS + 8);
        const float4 y3 = *reinterpret_cast<const float4*>(XT + kn * XTS + 12);
        const double scsn = SCl[2 * kn], sccn = SCl[2 * kn + 1];
        if (cb != 0xFFu) {
          const int k = (int)(cb & 0x7Fu);
          pi += r0 * (double)x2.y + r1 * (double)x2.z + r2 * (double)x2.w;  // X[9..11]: translation
          const double n0 = r0 * (double)x0.x + r1 * (double)x0.w + r2 * (double)x1.z;  // X[0], X[3], X[6]
          const double n1 = r0 * (double)x0.y + r1 * (double)x1.x + r2 * (double)x1.w;  // X[1], X[4], X[7]
          const double n2 = r0 * (double)x0.z + r1 * (double)x1.y + r2 * (double)x2.x;  // X[2], X[5], X[8]
          if ((revmask >> k) & 1u) {
            r0 = scc * n0 + scs * n1;
            r1 = scc * n1 - scs * n0;
            r2 = n2;
          } else {
            r0 = n0;
            r1 = n1;
            r2 = n2;
            pi += scs * n2;
          }
          if (cb & 0x80u) {  // this chain publishes the joint: each of its lanes its own coordinate
            AXl[k * 4 + ri] = (float)r2;
            OGl[k * 4 + ri] = (float)pi;
            const int fbe = __float_as_int(x3.x);
            const int fb = fbe & 0xFF, fe = fbe >> 8;
            for (int f = fb; f < fe; ++f) {
              const float4 fo = *reinterpret_cast<const float4*>(FO + f * 4);
              Pl[f * 3 + ri] = pi + r0 * (double)fo.x + r1 * (double)fo.y + r2 * (double)fo.z;
            }
          }
        }
        cb = cbn;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        scs = scsn; scc = sccn;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      return;
    }
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
    // the tables of step s + 1 are fetched while step s is computed (two LDS round trips per step off the chain) --
    // except in the three-waves-per-SIMD build of the 16-row grid, where the 20 extra live registers spill
    constexpr bool PREFETCH = MIMIC || NMAX > 16;
    unsigned cb = depth > 0 ? CH[l * 16] : 0xFFu;
    int kf = cb != 0xFFu ? (int)(cb & 0x7Fu) : 0;
    float4 x0 = *reinterpret_cast<const float4*>(XT + kf * XTS);
    float4 x1 = *reinterpret_cast<const float4*>(XT + kf * XTS + 4);
    float4 x2 = *reinterpret_cast<const float4*>(XT + kf * XTS + 8);
    float4 x3 = *reinterpret_cast<const float4*>(XT + kf * XTS + 12);
    double scs = SCl[2 * kf], scc = SCl[2 * kf + 1];
#pragma clang loop unroll(disable) vectorize(disable)
    for (int s = 0; s < depth; ++s) {
      unsigned cbn = 0xFFu;
      float4 y0, y1, y2, y3;
      double scsn, sccn;
      auto fetch_next = [&]() {
        cbn = (s + 1 < depth) ? CH[l * 16 + s + 1] : 0xFFu;
        const int kn = cbn != 0xFFu ? (int)(cbn & 0x7Fu) : 0;
        y0 = *reinterpret_cast<const float4*>(XT + kn * XTS);
        y1 = *reinterpret_cast<const float4*>(XT + kn * XTS + 4);
        y2 = *reinterpret_cast<const float4*>(XT + kn * XTS + 8);
        y3 = *reinterpret_cast<const float4*>(XT + kn * XTS + 12);
        scsn = SCl[2 * kn];
        sccn = SCl[2 * kn + 1];
      };
      if (PREFETCH) fetch_next();
      if (cb != 0xFFu) {
        const int k = (int)(cb & 0x7Fu);
        const double Xk[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
#pragma unroll
        for (int i = 0; i < 3; ++i) pp[i] += R[3 * i] * Xk[9] + R[3 * i + 1] * Xk[10] + R[3 * i + 2] * Xk[11];
        double Rn[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) Rn[3 * i + j] = R[3 * i] * Xk[j] + R[3 * i + 1] * Xk[3 + j] + R[3 * i + 2] * Xk[6 + j];
        if ((revmask >> k) & 1u) {
          const double sn = scs, cs = scc;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double c0 = Rn[3 * i], c1 = Rn[3 * i + 1];
            R[3 * i] = cs * c0 + sn * c1;
            R[3 * i + 1] = cs * c1 - sn * c0;
            R[3 * i + 2] = Rn[3 * i + 2];
          }
        } else {
          const double q = scs;
#pragma unroll
          for (int i = 0; i < 9; ++i) R[i] = Rn[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) pp[i] += q * Rn[3 * i + 2];
        }
        if (cb & 0x80u) {  // this lane publishes the joint
          *reinterpret_cast<float4*>(AXl + k * 4) = make_float4((float)R[2], (float)R[5], (float)R[8], 0.f);
          *reinterpret_cast<float4*>(OGl + k * 4) = make_float4((float)pp[0], (float)pp[1], (float)pp[2], 0.f);
          const int fbe = __float_as_int(x3.x);
          const int fb = fbe & 0xFF, fe = fbe >> 8;
          for (int f = fb; f < fe; ++f) {
            const float4 fo = *reinterpret_cast<const float4*>(FO + f * 4);
#pragma unroll
            for (int i = 0; i < 3; ++i)
              Pl[f * 3 + i] = pp[i] + R[3 * i] * (double)fo.x + R[3 * i + 1] * (double)fo.y + R[3 * i + 2] * (double)fo.z;
          }
        }
      }
      if (!PREFETCH) fetch_next();
      cb = cbn;
      x0 = y0; x1 = y1; x2 = y2; x3 = y3;
      scs = scsn; scc = sccn;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // ---- value / gradient (own joints) / Hessian grid at the kinematic state in LDS ------------------------------------
  float gnew[NJ2];
  wv2 Hn[NR][NP];
  // SPRINT assembles a model only at a point it has already decided to keep: the new Hessian is accumulated straight into the
  // accepted one's registers (36 fewer live registers at n = 24 than with a separate Hn; four frames per wave need both, the
  // decision comes after the assembly there)
  wv2 (&Hx)[NR][NP] = SPRINT ? Ha : Hn;
  // (1) lane t evaluates term t; returns F (identical in the 16 lanes of the row)
  auto terms = [&]() -> double {
    double Fv = 0;
    if (l < nt) {
      const int ft = my_ft, fo = my_fo;
      double rd[3];
      float tv[3], wgt;
      if (TGLDS) {
        const float4 tg = *reinterpret_cast<const float4*>(TGl + l * 4);
        tv[0] = tg.x; tv[1] = tg.y; tv[2] = tg.z; wgt = tg.w;
      } else if (TGREG) {
        tv[0] = tgt[0]; tv[1] = tgt[1]; tv[2] = tgt[2]; wgt = tgt[3];
      } else {
        int row = my_ref;
        asm volatile("" : "+v"(row));
        term_target(cold(), row, tv, wgt);  // (n = 32: every pass)
      }
      float ptf[3], pof[3] = {0, 0, 0};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double pt = Pl[ft * 3 + i];
        const double po = fo >= 0 ? Pl[fo * 3 + i] : 0.0;
        rd[i] = pt - po - (double)tv[i];
        ptf[i] = (float)pt;
        pof[i] = (float)po;
      }
      const double w = (double)kp.inv_norm * (double)wgt;
      float fvec[3], hw[3], kap = 0;
      if (per_coord) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double ee = rd[i], ae = fabs(ee);
          const bool quad = ae < beta;
          Fv += w * (quad ? 0.5 * ee * ee * ibeta : ae - 0.5 * beta);
          fvec[i] = (float)(w * (quad ? ee * ibeta : (ee > 0 ? 1.0 : -1.0)));
          hw[i] = (float)(w * (quad ? ibeta : (newton ? 0.0 : 1.0 / ae)));
        }
      } else {
        const double d2 = rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2];
        const double dd = sqrt(d2);
        const bool quad = dd < beta;
        Fv += w * (quad ? 0.5 * d2 * ibeta : dd - 0.5 * beta);
        const double id = quad ? ibeta : 1.0 / dd;
        const double psi = w * id;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          fvec[i] = (float)(psi * rd[i]);
          hw[i] = (float)psi;
        }
        // (round 6) the Huber curvature psi (I - r r^T / d^2) of the linear regime is psi P with P a PROJECTION (P = P^T P):
        // J^T (psi P) J = (sqrt(psi) P J)^T (sqrt(psi) P J) -- three weighted rows (c_i - r^_i (r^ . c)) instead of three rows plus
        // a fourth, subtracted rank-one row: a quarter fewer outer products in the term loop.  rq = r^ there, 0 in the
        // quadratic regime (P = I).
        kap = quad ? 0.f : (float)id;
      }
      float* T = TBl + l * 16;
      // the Hessian is accumulated as sum_k (sqrt(w_k) J_k) (sqrt(w_k) J_k)^T: the square roots of the row weights travel with
      // the term, and so does rq (the unit residual where the Huber loss is linear, else 0): see above
      *reinterpret_cast<float4*>(T) = make_float4((float)rd[0] * kap, (float)rd[1] * kap, (float)rd[2] * kap, 0.f);  // rq (kap: 1 / d or 0)
      *reinterpret_cast<float4*>(T + 4) = make_float4(fvec[0], fvec[1], fvec[2], __fsqrt_rn(hw[0]));
      *reinterpret_cast<float4*>(T + 8) = make_float4(ptf[0], ptf[1], ptf[2], __fsqrt_rn(hw[1]));
      *reinterpret_cast<float4*>(T + 12) = make_float4(pof[0], pof[1], pof[2], __fsqrt_rn(hw[2]));
    }
    // regulariser of the own joints
#pragma unroll
    for (int s = 0; s < NJ2; ++s)
      if (jopt[s]) {
        const double dx = (double)xj[s] - (double)XLl[jo_[s]];
        Fv += (double)delta * dx * dx;
      }
    Fv = row_sum64(Fv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return Fv;
  };
  // gradient (own joints) and Hessian grid of the data term at the kinematic state / term blocks the slot pointers address
  auto model_rest = [&]() {
    // (2) own joints' axes / origins; accumulators of the pass: data-term gradient and second-order vector
    constexpr int NCOL = MIMIC ? FAM : NJ2;  // joints whose columns this lane forms
    float jax[NCOL][3], jog[NCOL][3], jcf[NCOL][3];
#pragma unroll
    for (int s = 0; s < NCOL; ++s) {
      const int kj = MIMIC ? (famk[s] >= 0 ? famk[s] : 0) : (jin[s] ? jo_[s] : 0);
      const float4 av = *reinterpret_cast<const float4*>(AXl + kj * 4);
      const float4 ov = *reinterpret_cast<const float4*>(OGl + kj * 4);
      jax[s][0] = av.x; jax[s][1] = av.y; jax[s][2] = av.z;
      jog[s][0] = ov.x; jog[s][1] = ov.y; jog[s][2] = ov.z;
      jcf[s][0] = 0; jcf[s][1] = 0; jcf[s][2] = 0;
    }
#pragma unroll
    for (int s = 0; s < NJ2; ++s) gnew[s] = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int j = 0; j < NP; ++j) Hx[i][j] = wv2{0.f, 0.f};
    // packed views of the two joint slots (NJ2 == 2, joint-space grids): axes / origins, accumulators, per-lane masks
    constexpr int S1 = (MIMIC || NJ2 == 2) ? 1 : 0;  // (MIMIC: family joints 0 and 1 of the lane's variable)
    wv2 jax2[3], jog2[3], jcf2[3], gnew2 = wv2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      jax2[i] = wv2{jax[0][i], jax[S1][i]};
      jog2[i] = wv2{jog[0][i], jog[S1][i]};
      jcf2[i] = wv2{0.f, 0.f};
    }
    const wv2 jm2 = wv2{jopt[0] ? 1.f : 0.f, jopt[NJ2 - 1] ? 1.f : 0.f};  // optimised joints only
    // MIMIC: family joints 0 / 1 of this lane's variable -- present?, their bit positions, dq/dx, revolute?
    const int fk2x = MIMIC ? (famk[0] >= 0 ? famk[0] : 0) : 0, fk2y = MIMIC ? (famk[FAM > 1 ? 1 : 0] >= 0 ? famk[FAM > 1 ? 1 : 0] : 0) : 0;
    const wv2 fon2 = MIMIC ? wv2{famk[0] >= 0 ? 1.f : 0.f, (fam_max > 1 && famk[FAM > 1 ? 1 : 0] >= 0) ? 1.f : 0.f} : wv2{0.f, 0.f};
    const wv2 fm2 = MIMIC ? wv2{famm[0], famm[FAM > 1 ? 1 : 0]} : wv2{0.f, 0.f};
    const wv2 jr2 = MIMIC ? wv2{(float)((revmask >> fk2x) & 1u), (float)((revmask >> fk2y) & 1u)}
                          : wv2{jrev[0] ? 1.f : 0.f, jrev[NJ2 - 1] ? 1.f : 0.f};  // revolute joints

    // (3) terms in sequence: lane l forms the Jacobian columns of its joints (l, l + 16), accumulates their gradient
    // entries and second-order vectors and publishes the term's four weighted Jacobian rows; then every lane adds the
    // rows' outer products to its Hessian entries.
    // (Measured alternative: one lane per joint ON THE TERM'S CHAINS from a host-built list -- 11-12 of a Shadow hand's
    // 24 joints move a DexPilot pair -- with the owner lanes collecting gradient / second-order entries through LDS:
    // 37 % fewer VALU instructions in this loop, the same 1.9 ms for Shadow DexPilot and 10 % slower for the 16-joint
    // hands; the pass is bound by LDS round trips, not by VALU issue.)
    constexpr int nrow = 3;  // weighted rows per term (round 6: the Huber rank-one correction is folded into the three rows)
    // columns of term t -> gradient / second-order accumulators and the term's weighted Jacobian rows in buffer `buf`
    auto publish = [&](int t, bool on) {  // on: SPRINT rows beyond the last term of their share contribute zero columns
      float* JRw = JRl;
      const float* T = TBl + t * 16;
      const float4 t0 = *reinterpret_cast<const float4*>(T);
      const float4 t1 = *reinterpret_cast<const float4*>(T + 4);
      const float4 t2 = *reinterpret_cast<const float4*>(T + 8);
      const float4 t3 = *reinterpret_cast<const float4*>(T + 12);
      const uint32_t mt = TMw[t * TMS], mo = TMw[t * TMS + 1];
      if (MIMIC) {
        // the variable's column is the vmul-weighted sum over its joint family (kinematics_adaptor.py:102-113).  The
        // first two family joints (the variable's own joint and its first follower: all of Ability / Inspire, most of
        // SVH) are formed together in packed float32 arithmetic, like the two joint slots of the joint-space grids; a
        // third family joint takes the scalar path.
        float c0, c1, c2;
        {
          const wv2 fonv = (SPRINT && !on) ? wv2{0.f, 0.f} : fon2;
          const wv2 ft = wv2{(float)((mt >> fk2x) & 1u), (float)((mt >> fk2y) & 1u)} * fonv;
          const wv2 fo = wv2{(float)((mo >> fk2x) & 1u), (float)((mo >> fk2y) & 1u)} * fonv;
          const wv2 sg = ft - fo;
          wv2 v[3], d[3];
          v[0] = ft * t2.x - fo * t3.x - sg * jog2[0];
          v[1] = ft * t2.y - fo * t3.y - sg * jog2[1];
          v[2] = ft * t2.z - fo * t3.z - sg * jog2[2];
          d[0] = jax2[1] * v[2] - jax2[2] * v[1];
          d[1] = jax2[2] * v[0] - jax2[0] * v[2];
          d[2] = jax2[0] * v[1] - jax2[1] * v[0];
          if (any_prismatic) {
            const wv2 w = (wv2{1.f, 1.f} - jr2) * sg;
#pragma unroll
            for (int i = 0; i < 3; ++i) d[i] = d[i] * jr2 + jax2[i] * w;
          }
          jcf2[0] += d[1] * t1.z - d[2] * t1.y;
          jcf2[1] += d[2] * t1.x - d[0] * t1.z;
          jcf2[2] += d[0] * t1.y - d[1] * t1.x;
          const wv2 m0 = fm2 * d[0], m1 = fm2 * d[1], m2 = fm2 * d[2];
          c0 = m0.x + m0.y;
          c1 = m1.x + m1.y;
          c2 = m2.x + m2.y;
        }
#pragma unroll
        for (int e = 2; e < FAM; ++e) {
          if (e < fam_max) {
            const int k = famk[e] >= 0 ? famk[e] : 0;
            const bool fam_on = famk[e] >= 0 && (!SPRINT || on);
            const bool in_t = fam_on && ((mt >> k) & 1u), in_o = fam_on && ((mo >> k) & 1u);
            float d0 = 0, d1 = 0, d2 = 0;
            if (in_t || in_o) {
              if ((revmask >> k) & 1u) {
                float v0 = 0, v1 = 0, v2 = 0;
                if (in_t) { v0 += t2.x - jog[e][0]; v1 += t2.y - jog[e][1]; v2 += t2.z - jog[e][2]; }
                if (in_o) { v0 -= t3.x - jog[e][0]; v1 -= t3.y - jog[e][1]; v2 -= t3.z - jog[e][2]; }
                d0 = jax[e][1] * v2 - jax[e][2] * v1;
                d1 = jax[e][2] * v0 - jax[e][0] * v2;
                d2 = jax[e][0] * v1 - jax[e][1] * v0;
              } else {
                const float sg = (in_t ? 1.f : 0.f) - (in_o ? 1.f : 0.f);
                d0 = sg * jax[e][0]; d1 = sg * jax[e][1]; d2 = sg * jax[e][2];
              }
              jcf[e][0] += d1 * t1.z - d2 * t1.y;
              jcf[e][1] += d2 * t1.x - d0 * t1.z;
              jcf[e][2] += d0 * t1.y - d1 * t1.x;
            }
            c0 += famm[e] * d0; c1 += famm[e] * d1; c2 += famm[e] * d2;
          }
        }
        gnew[0] += c0 * t1.x + c1 * t1.y + c2 * t1.z;
        const int pos = (l & 3) * NRP + (l >> 2);
        if (!per_coord) {
          asm volatile("");
          const float u = c0 * t0.x + c1 * t0.y + c2 * t0.z;
          c0 -= t0.x * u; c1 -= t0.y * u; c2 -= t0.z * u;
        }
        JRw[0 * 4 * NRP + pos] = c0 * t1.w;
        JRw[1 * 4 * NRP + pos] = c1 * t2.w;
        JRw[2 * 4 * NRP + pos] = c2 * t3.w;
      } else if (NJ2 == 2) {
        // both joint slots of the lane (l, l + 16) at once in packed float32 arithmetic (v_pk_*): the two columns are
        // the same formula on different operands, and a pass is bound by the number of VALU instructions issued.
        //   v = in_t (p_task - o) - in_o (p_origin - o);  column = a x v (revolute)  |  (in_t - in_o) a (prismatic)
        const wv2 jmv = (SPRINT && !on) ? wv2{0.f, 0.f} : jm2;
        const wv2 ft = wv2{(float)((mt >> l) & 1u), (float)((mt >> (l + 16)) & 1u)} * jmv;
        const wv2 fo = wv2{(float)((mo >> l) & 1u), (float)((mo >> (l + 16)) & 1u)} * jmv;
        const wv2 sg = ft - fo;
        wv2 v[3], c[3];
        v[0] = ft * t2.x - fo * t3.x - sg * jog2[0];
        v[1] = ft * t2.y - fo * t3.y - sg * jog2[1];
        v[2] = ft * t2.z - fo * t3.z - sg * jog2[2];
        c[0] = jax2[1] * v[2] - jax2[2] * v[1];
        c[1] = jax2[2] * v[0] - jax2[0] * v[2];
        c[2] = jax2[0] * v[1] - jax2[1] * v[0];
        if (any_prismatic) {  // wave-uniform: only models with translation joints (the dummy free base) pay for the blend
          const wv2 w = (wv2{1.f, 1.f} - jr2) * sg;
#pragma unroll
          for (int i = 0; i < 3; ++i) c[i] = c[i] * jr2 + jax2[i] * w;
        }
        gnew2 += c[0] * t1.x + c[1] * t1.y + c[2] * t1.z;
        jcf2[0] += c[1] * t1.z - c[2] * t1.y;
        jcf2[1] += c[2] * t1.x - c[0] * t1.z;
        jcf2[2] += c[0] * t1.y - c[1] * t1.x;
        wv2 r0 = c[0] * t1.w, r1 = c[1] * t2.w, r2 = c[2] * t3.w;
        if (!per_coord) {  // (wave-uniform; the empty asm keeps it a scalar BRANCH: if-converted, the position models paid for
                           // both forms and six selects per term -- LEAP position + 3 % in the same-box A/B)
          asm volatile("");
          const wv2 u = c[0] * t0.x + c[1] * t0.y + c[2] * t0.z;
          r0 = (c[0] - u * t0.x) * t1.w;
          r1 = (c[1] - u * t0.y) * t2.w;
          r2 = (c[2] - u * t0.z) * t3.w;
        }
        // row k of the term's Jacobian^T: position (k mod 4) * NRP + k / 4 of each of the four rows (k = l, l + 16)
        const int pos0 = (l & 3) * NRP + (l >> 2), pos1 = pos0 + 4;
        JRw[0 * 4 * NRP + pos0] = r0.x; JRw[0 * 4 * NRP + pos1] = r0.y;
        JRw[1 * 4 * NRP + pos0] = r1.x; JRw[1 * 4 * NRP + pos1] = r1.y;
        JRw[2 * 4 * NRP + pos0] = r2.x; JRw[2 * 4 * NRP + pos1] = r2.y;
      } else {
#pragma unroll
        for (int s = 0; s < NJ2; ++s) {
          const int k = jo_[s];
          const bool in_t = (mt >> k) & 1u, in_o = (mo >> k) & 1u;
          float c0 = 0, c1 = 0, c2 = 0;
          if (jopt[s] && (in_t || in_o) && (!SPRINT || on)) {
            if (jrev[s]) {
              float v0 = 0, v1 = 0, v2 = 0;
              if (in_t) { v0 += t2.x - jog[s][0]; v1 += t2.y - jog[s][1]; v2 += t2.z - jog[s][2]; }
              if (in_o) { v0 -= t3.x - jog[s][0]; v1 -= t3.y - jog[s][1]; v2 -= t3.z - jog[s][2]; }
              c0 = jax[s][1] * v2 - jax[s][2] * v1;
              c1 = jax[s][2] * v0 - jax[s][0] * v2;
              c2 = jax[s][0] * v1 - jax[s][1] * v0;
            } else {
              const float sg = (in_t ? 1.f : 0.f) - (in_o ? 1.f : 0.f);
              c0 = sg * jax[s][0]; c1 = sg * jax[s][1]; c2 = sg * jax[s][2];
            }
            gnew[s] += c0 * t1.x + c1 * t1.y + c2 * t1.z;
            jcf[s][0] += c1 * t1.z - c2 * t1.y;
            jcf[s][1] += c2 * t1.x - c0 * t1.z;
            jcf[s][2] += c0 * t1.y - c1 * t1.x;
          }
          // row k of the term's Jacobian^T: position (k mod 4) * NRP + k / 4 of each of the four rows
          const int pos = (k & 3) * NRP + (k >> 2);
          if (!per_coord) {
            asm volatile("");
            const float u = c0 * t0.x + c1 * t0.y + c2 * t0.z;
            c0 -= t0.x * u; c1 -= t0.y * u; c2 -= t0.z * u;
          }
          JRw[0 * 4 * NRP + pos] = c0 * t1.w;
          JRw[1 * 4 * NRP + pos] = c1 * t2.w;
          JRw[2 * 4 * NRP + pos] = c2 * t3.w;
        }
      }
    };
    // every lane adds the outer products of the rows in buffer `buf` to its Hessian entries
    auto outer = [&]() {
      const float* JRr = JRl;
#pragma unroll
      for (int kk = 0; kk < nrow; ++kk) {
        {
          float jr[NRP];
          wv2 jc[NRP / 2];
#pragma unroll
          for (int i = 0; i < NRP; i += 4) {
            if (i < NR) {
              const float4 rv = *reinterpret_cast<const float4*>(JRr + kk * 4 * NRP + a * NRP + i);
              const float4 cv = *reinterpret_cast<const float4*>(JRr + kk * 4 * NRP + b * NRP + i);
              jr[i] = rv.x; jr[i + 1] = rv.y; jr[i + 2] = rv.z; jr[i + 3] = rv.w;
              jc[i / 2] = wv2{cv.x, cv.y};
              jc[i / 2 + 1] = wv2{cv.z, cv.w};
            }
          }
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            const wv2 rr = wv2{jr[i], jr[i]};
#pragma unroll
            for (int jj = 0; jj <= i / 2; ++jj) Hx[i][jj] = __builtin_elementwise_fma(rr, jc[jj], Hx[i][jj]);
          }
        }
      }
    };
    // (Measured and dropped: publishing term t + 1 into a second row buffer while term t's outer products run -- one
    // barrier per term, no exposed LDS write -> read latency -- changed nothing: tools/prof_wide_stages.sh shows the same
    // cycles per pass for a lone wave and for two waves per SIMD, i.e. a pass is bound by the wave's own instruction
    // issue (~8 cycles per dependent VALU instruction), not by LDS round trips.  Fewer instructions is what helps.)
    if (SPRINT) {
      const int ntq = (nt + 3) >> 2;  // row s: terms s, s + 4, ...
#pragma clang loop unroll(disable) vectorize(disable)
      for (int tq = 0; tq < ntq; ++tq) {
        const int t = 4 * tq + slot;
        const bool on = t < nt;
        publish(on ? t : 0, on);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        outer();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    } else {
#pragma clang loop unroll(disable) vectorize(disable)
      for (int t = 0; t < nt; ++t) {
        publish(t, true);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        outer();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }

    if (!MIMIC && NJ2 == 2) {
      gnew[0] = gnew2.x;
      gnew[NJ2 - 1] = gnew2.y;
    }
    if (MIMIC || NJ2 == 2) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        jcf[0][i] += jcf2[i].x;
        jcf[S1][i] += jcf2[i].y;
      }
    }
    if (SPRINT) {
      // the rows' partial sums -> every row: v + v(lane ^ 16), then + (lane ^ 32).  Both partners of an exchange add the same
      // two numbers, so the four rows end with identical bits (the redundant stages after this stay in lockstep).
      const int x16 = (lane ^ 16) << 2, x32 = (lane ^ 32) << 2;
      auto xsum = [&](float v) -> float {
        v += __int_as_float(__builtin_amdgcn_ds_bpermute(x16, __float_as_int(v)));
        v += __int_as_float(__builtin_amdgcn_ds_bpermute(x32, __float_as_int(v)));
        return v;
      };
#pragma unroll
      for (int s2 = 0; s2 < NJ2; ++s2) gnew[s2] = xsum(gnew[s2]);
#pragma unroll
      for (int s2 = 0; s2 < NCOL; ++s2)
#pragma unroll
        for (int i = 0; i < 3; ++i) jcf[s2][i] = xsum(jcf[s2][i]);
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j <= i / 2; ++j) {
          Hx[i][j].x = xsum(Hx[i][j].x);
          Hx[i][j].y = xsum(Hx[i][j].y);
        }
    }
    WPROF_STAGE(3)
    // (4) second-order kinematic term: H[r][c] += a_c . CF_r for every revolute ancestor-or-self c of r
    if (newton && MIMIC) {
      // H[v][w] += sum over joint pairs (k in family v, j a revolute ancestor-or-self of k in family w) of
      // m_k m_j a_j . CF_k (twice when both are followers of one variable): the host lists the pairs per owner lane of
      // the target entry (dexr_api.hip: build_wide_tables), so a lane only adds into its own entries -- through private
      // LDS slots, a register file has no run-time index
#pragma unroll
      for (int e = 0; e < FAM; ++e)
        if (famk[e] >= 0) *reinterpret_cast<float4*>(CFl + famk[e] * 4) = make_float4(jcf[e][0], jcf[e][1], jcf[e][2], 0.f);
      float* HBl = TBl + l * 12;  // the term block is free now
      *reinterpret_cast<float4*>(HBl) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(HBl + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(HBl + 8) = make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int p0 = POFF[l], p1 = POFF[l + 1];
      const int pair_max = wt.pair_max;
#pragma clang loop unroll(disable) vectorize(disable)
      for (int it = 0; it < pair_max; ++it) {
        const int pp = p0 + it;
        if (pp < p1) {
          const uint32_t w = PLw[pp];
          const int k = (int)(w & 31u), j = (int)((w >> 5) & 31u), e = (int)((w >> 10) & 15u);
          const float4 aj = *reinterpret_cast<const float4*>(AXl + j * 4);
          const float4 cf = *reinterpret_cast<const float4*>(CFl + k * 4);
          float val = XT[k * XTS + 14] * XT[j * XTS + 14] * (aj.x * cf.x + aj.y * cf.y + aj.z * cf.z);
          if ((w >> 14) & 1u) val += val;
          HBl[e] += val;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          const float add = HBl[i * (i + 1) / 2 + j];
          if (j & 1) Hx[i][j / 2].y += add; else Hx[i][j / 2].x += add;
        }
    }
    if (newton && !MIMIC) {
#pragma unroll
      for (int s = 0; s < NJ2; ++s)
        if (jin[s]) *reinterpret_cast<float4*>(CFl + jo_[s] * 4) = make_float4(jcf[s][0], jcf[s][1], jcf[s][2], 0.f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float4 axc[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) axc[j] = *reinterpret_cast<const float4*>(AXl + (4 * j + b) * 4);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const float4 cf = *reinterpret_cast<const float4*>(CFl + (4 * i + a) * 4);
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          const float v = axc[j].x * cf.x + axc[j].y * cf.y + axc[j].z * cf.z;
          const float add = ((ANCw[4 * i + a] >> (4 * j + b)) & 1u) ? v : 0.f;
          if (j & 1) Hx[i][j / 2].y += add; else Hx[i][j / 2].x += add;
        }
      }
    }
  };
  auto assemble = [&]() -> double {
    const double Fv = terms();
    WPROF_STAGE(2)
    model_rest();
    return Fv;
  };

  // ---- distributed Cholesky + triangular solves of (H_acc restricted to the free set + damping) d = -g -------------
  float dstep[NJ2];  // own joints' entries of the step
  float hdmean = 0.f;  // mean diagonal of the free block of the accepted Hessian (MODCHOL: scale of the damping jump)
  auto factor_and_solve = [&](uint32_t freemask, float lam) -> bool {
    bool ok = true;
    if (MODCHOL) {
      float hds = 0.f;
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (a == b && ((freemask >> (4 * i + a)) & 1u)) hds += (i & 1) ? Ha[i][i / 2].y : Ha[i][i / 2].x;
      const int nfree = __popc(freemask);
      hdmean = row_sum(hds) / (float)(nfree > 0 ? nfree : 1);
    }
    wv2 Hw[NR][NP];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = 4 * i + a;
      const bool fr = (freemask >> r) & 1u;
#pragma unroll
      for (int j = 0; j <= (i | 1); ++j) {
        const int c = 4 * j + b;
        const bool fc = (freemask >> c) & 1u;
        float v = (fr && fc && j <= i) ? ((j & 1) ? Ha[i][j / 2].y : Ha[i][j / 2].x) : 0.f;
        if (i == j && a == b) v = fr ? v + 2.f * delta + lam : 1.f;
        if (j & 1) Hw[i][j / 2].y = v; else Hw[i][j / 2].x = v;
      }
    }
    auto hw_get = [&](int i, int j) -> float { return (j & 1) ? Hw[i][j / 2].y : Hw[i][j / 2].x; };
    // The right-hand side rides along as row NMAX of the matrix (held by quad 0, class a = 0): the factorisation's
    // trailing updates then ARE the forward substitution -- after the last step that row holds y = L^-1 rhs.  Saves a
    // 24-step dependent chain of quad sums and row broadcasts per pass.
    wv2 Hy[NP];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int c = 4 * j + b;
      const float v = (a == 0 && ((freemask >> c) & 1u)) ? -GVl[c] : 0.f;
      if (j & 1) Hy[j / 2].y = v; else Hy[j / 2].x = v;
    }
    auto hy_get = [&](int j) -> float { return (j & 1) ? Hy[j / 2].y : Hy[j / 2].x; };
    float ivc[NR];  // 1 / L[c][c] of column 4 j + b
#pragma unroll
    for (int i = 0; i < NR; ++i) ivc[i] = 0.f;
#pragma unroll
    for (int jc = 0; jc < NMAX; ++jc) {
      const int ja = jc & 3, jo = jc >> 2;
      // pivot: lane (ja, ja) -> its quad (DPP) -> the row (mask + stride-4 sum)
      const float own_d = hw_get(jo, jo);
      const float qd = ja == 0 ? wquad_bcast<0>(own_d) : ja == 1 ? wquad_bcast<1>(own_d) : ja == 2 ? wquad_bcast<2>(own_d) : wquad_bcast<3>(own_d);
      float dj = stride4_sum(a == ja ? qd : 0.f);
      if (MODCHOL) {
        if (!(dj > 1e-6f * (2.f * delta + lam))) { ok = false; dj = fmaxf(fabsf(dj), 2.f * delta + lam); }
      } else {
        if (!(dj > 1e-30f)) { ok = false; dj = 1.f; }
      }
      const float iv = __frsqrt_rn(dj);
      ivc[jo] = (b == ja) ? iv : ivc[jo];
      // scale column jc (lanes of column class ja), rows below the pivot; then send it to the grid: row side from
      // lane (a, ja) (own quad, DPP), column side from lane (b, ja) (ds_bpermute)
      float Lr[NR];
      wv2 Lc[NP];
#pragma unroll
      for (int jj = 0; jj < NP; ++jj) Lc[jj] = wv2{0.f, 0.f};
#pragma unroll
      for (int i = jo; i < NR; ++i) {
        const bool below = (i > jo) || (a > ja);
        const float cur = hw_get(i, jo);
        const float own = (b == ja && below) ? cur * iv : cur;
        if (jo & 1) Hw[i][jo / 2].y = own; else Hw[i][jo / 2].x = own;
        float vr = ja == 0 ? wquad_bcast<0>(own) : ja == 1 ? wquad_bcast<1>(own) : ja == 2 ? wquad_bcast<2>(own) : wquad_bcast<3>(own);
        float vc = __int_as_float(__builtin_amdgcn_ds_bpermute(rowbase4 + 16 * b + 4 * ja, __float_as_int(own)));
        if (i == jo) {
          vr = (a > ja) ? vr : 0.f;
          vc = (b > ja) ? vc : 0.f;
        }
        Lr[i] = vr;
        if (i & 1) Lc[i / 2].y = vc; else Lc[i / 2].x = vc;
      }
      // the pivot column itself (local column jo on lanes with b == ja) sees Lc[jo] = 0 there: it stays as scaled
#pragma unroll
      for (int i = jo; i < NR; ++i) {
        const wv2 rr = wv2{-Lr[i], -Lr[i]};
#pragma unroll
        for (int jj = jo / 2; jj <= i / 2; ++jj) Hw[i][jj] = __builtin_elementwise_fma(rr, Lc[jj], Hw[i][jj]);
      }
      {  // the right-hand-side row
        const float cur = hy_get(jo);
        const float own = (b == ja) ? cur * iv : cur;
        if (jo & 1) Hy[jo / 2].y = own; else Hy[jo / 2].x = own;
        const float ly = ja == 0 ? wquad_bcast<0>(own) : ja == 1 ? wquad_bcast<1>(own) : ja == 2 ? wquad_bcast<2>(own) : wquad_bcast<3>(own);
        const wv2 rr = wv2{-ly, -ly};
#pragma unroll
        for (int jj = jo / 2; jj < NP; ++jj) Hy[jj] = __builtin_elementwise_fma(rr, Lc[jj], Hy[jj]);
      }
    }
    // y[4 j + b] from quad 0 to the lanes of column class b of every quad
    float yb[NR], da[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      yb[j] = stride4_sum(a == 0 ? hy_get(j) : 0.f);
      da[j] = 0;
    }
    // backward: L^T d = y; da[i] = d[4 i + a]
#pragma unroll
    for (int s = 0; s < NJ2; ++s) dstep[s] = 0;
#pragma unroll
    for (int r = NMAX - 1; r >= 0; --r) {
      const int rb = r & 3, ro = r >> 2;
      float part = 0;
#pragma unroll
      for (int i = ro; i < NR; ++i) part += hw_get(i, ro) * da[i];  // rows at or above r meet da = 0
      const float sum = stride4_sum(part);                      // valid on lanes of column class rb
      const float dr_local = (yb[ro] - sum) * ivc[ro];
      const float dr = rb == 0 ? wquad_bcast<0>(dr_local) : rb == 1 ? wquad_bcast<1>(dr_local)
                     : rb == 2 ? wquad_bcast<2>(dr_local) : wquad_bcast<3>(dr_local);  // lane b = rb of every quad
      da[ro] = (a == rb) ? dr : da[ro];
#pragma unroll
      for (int s = 0; s < NJ2; ++s) dstep[s] = (jo_[s] == r) ? dr : dstep[s];
    }
    return ok;
  };

  // ---- projected Levenberg-Marquardt / Newton ------------------------------------------------------------------------
  float lam = kp.lam0, nu = 2.f, sprev = 1e30f, keff = 0.f;
  WDIAG(int d_nrej = 0; int d_ncap = 0; int d_nfail = 0;)
  bool done = true, pending = false;
  int status = ST_MAXITER, my_iters = 0, blind = 0, nrej = 0;  // nrej: rejections of this solve (bounds the fast damping decay)
  double F = 0;
  float smax = 0, pred = 0;
  bool ok = true;
  unsigned pool_next = (unsigned)((tile * FPW < (int64_t)kp.q0 && tile * FPW < nB) ? tile * FPW : 0);
  unsigned pool_end = (unsigned)((tile * FPW < (int64_t)kp.q0 && tile * FPW < nB) ? ((tile * FPW + FPW < nB) ? tile * FPW + FPW : nB) : 0);
  bool dry = false;
  unsigned* queue = kp.queue + comp;
  auto reset_state = [&]() {
    done = false;
    pending = false;
    lam = kp.lam0;
    nu = 2.f;
    sprev = 1e30f;
    keff = 0.f;
    nrej = 0;
    WDIAG(d_nrej = 0; d_ncap = 0; d_nfail = 0;)
    status = ST_MAXITER;
    my_iters = kp.iters_base;
    blind = 0;
    F = 0;
    smax = 0;
    pred = 0;
    ok = true;
  };
  float screen_acc = 0.f;  // screening launch: sum of F(x0) over the frames of this wave (lane 0 of each row)
  // SPRINT: the rows' damping multipliers (the ladder; all 1: the rows are copies) -- read once, selected by row where needed
  const float mu_0 = SPRINT ? kp.sprint_mu[0] : 1.f, mu_1 = SPRINT ? kp.sprint_mu[1] : 1.f, mu_2 = SPRINT ? kp.sprint_mu[2] : 1.f,
              mu_3 = SPRINT ? kp.sprint_mu[3] : 1.f;
  const float mu_own = slot == 0 ? mu_0 : slot == 1 ? mu_1 : slot == 2 ? mu_2 : mu_3;
  for (;;) {
    // the lane's grid coordinates, made opaque once per pass: predicates on them (a == 2, b <= a, ...) are then recomputed
    // where they are used (one v_cmp) instead of being hoisted out of the loop as 64-bit lane masks -- dozens of SGPR pairs
    // that do not fit and come back from VGPR lanes with two v_readlane + a hazard nop each
    asm volatile("" : "+v"(l), "+v"(a), "+v"(b));
    WPROF_START();
    // (0) hand frames to idle rows
    const unsigned long long want = __ballot(!active);
    if (want != 0ull) {
      if (pool_next >= pool_end && !dry) {
        // take exactly as many frames as there are idle rows: a frame parked in this wave's pool while its other rows
        // are busy would start late (near the end of the queue other waves' rows are idle by then)
        const unsigned nwant = SPRINT ? 1u : (unsigned)__popcll(want) >> 4;  // (SPRINT: the four rows are idle together)
        unsigned base = 0;
        // (static tiles cover the batch: nothing to draw, the counter -- which such launches do not reset -- is not touched)
        const bool no_queue = (int64_t)kp.q0 >= kp.B;
        if (!no_queue && lane == 0) base = atomicAdd(queue, nwant);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + kp.q0;
        if (no_queue || (int64_t)base >= nB) {
          dry = true;
        } else {
          pool_next = base;
          pool_end = (unsigned)(((int64_t)base + nwant < nB) ? base + nwant : nB);
        }
      }
      if (pool_next < pool_end) {
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
        const unsigned cand = SPRINT ? pool_next : pool_next + ((below - (unsigned)l) >> 4);  // (SPRINT: one frame, every row)
        const bool got = !active && cand < pool_end;
        pool_next += SPRINT ? (__ballot(got) != 0ull ? 1u : 0u) : (unsigned)__popcll(__ballot(got)) >> 4;
        if (got) {
          load_frame((int64_t)cand, 0);
          load_target();
          active = true;
          reset_state();
        }
      }
    }
    if (!__any(active)) {
      if (dry && pool_next >= pool_end) break;
      continue;
    }
    WPROF_STAGE(0)
    fk();
    WPROF_STAGE(1)
    if (kp.screen) {
      // SCREENING launch (longest-first ordering, dexr_api.hip: launch_wide): F at the start point is all that is
      // wanted of a frame -- large values mark the frames that will need many passes (DexPilot models: the top 5 % by
      // F(x0) hold 91 % of the frames with >= 15 iterations)
      const double F0 = terms();
      if (active) {
        if (l == 0) kp.screen[f_lrow()] = (float)F0;
        screen_acc += (l == 0) ? (float)F0 : 0.f;
        active = false;
        done = true;
      }
      continue;
    }
    // SPRINT evaluates the VALUE first and the model only at the point it keeps: with a ladder of damping values the four rows
    // hold four different trial points (own kinematic state and term blocks in the row's own frame slot), the best acceptable one
    // wins, and the split term loop then runs on the winner's slot; a pass whose steps are all rejected assembles nothing.
    double Fe = SPRINT ? terms() : assemble();
    WPROF_STAGE(4)
    if (!done) {
      bool take = false;
      bool sprint_floor = false;  // SPRINT: see below
      int wrow = slot;  // SPRINT: the row whose trial point is kept
      if (!pending) {
        F = Fe;
        take = true;
      } else {
        // (float: the float64 product kept (double) floor_scale alive across the pass loop, in scratch; pred is a float anyway)
        const float noise = kp.floor_scale * fabsf((float)F);
        // (bitwise, not short-circuit: `&&` / `||` on lane-varying conditions compile to nested exec-mask branches)
        bool finite = (bool)((int)(Fe == Fe) & (int)(smax == smax) & (int)(fabs(Fe) < 1e30));
        bool below_floor = (bool)((int)ok & (int)finite & (int)(pred <= noise) & (int)(smax < 1e-2f));
        bool accept = (bool)((int)(ok || MODCHOL) & (int)finite & ((int)(Fe <= F) | (int)below_floor));
        if (SPRINT) {
          // the four rows' candidates: the acceptable one with the lowest value wins (ties: the least damped); its trial point,
          // value and step statistics replace every row's own, and `lam` becomes the damping it was computed with.  No row
          // acceptable: a rejection from the LARGEST damping tried (row 3 holds it, and the curvature along its step).
          const double score = accept ? Fe : 1e300;
          double sc[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sc[r] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(score), 16 * r), __builtin_amdgcn_readlane(__double2loint(score), 16 * r));
          int w = 0;
          double best = sc[0];
#pragma unroll
          for (int r = 1; r < 4; ++r)
            if (sc[r] < best) { best = sc[r]; w = r; }
          const bool any_ok = best < 1e300;
          // "every row's step was tiny, verified and finite": the rejection-at-the-floor exit needs it of all four
          const bool small_own = (bool)((int)(ok || MODCHOL) & (int)finite & (int)(smax < kp.tol));
          const bool all_small = __ballot(small_own) == ~0ull;
          const bool all_finite = __ballot(finite) == ~0ull;
          // (round 6) every row's step is below the tolerance and the least damped row's damping is small: the frame sits at the
          // rounding floor of its float32 coordinates -- the less damped rows step one ulp uphill and are rejected, a heavily
          // damped row "steps" zero ulps and is accepted, and its damping (x 3 or x 30, above lam_ok) kept the tiny-step exit
          // below from firing, pass after pass up to max_iter (12 of 512 reachable-target frames of the SVH position model once
          // the ladder stopped taking blind steps).  Same rule as the four-per-wave iteration's "a rejected step below tol ends
          // the solve at the rounding floor".
          sprint_floor = (bool)((int)all_small & (int)(lam * mu_0 <= fmaxf(2.f * delta, 10.f * kp.lam0)));
          if (any_ok) {
            const int src = (16 * w + l) << 2;
#pragma unroll
            for (int s2 = 0; s2 < NJ2; ++s2) xj[s2] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(xj[s2])));
            pred = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pred)));
            smax = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(smax)));
            const int flags = __builtin_amdgcn_ds_bpermute(src, (int)ok | ((int)below_floor << 1));
            ok = (bool)(flags & 1);
            below_floor = (bool)((flags >> 1) & 1);
            Fe = best;
            finite = true;
            accept = true;
            lam = lam * (w == 0 ? mu_0 : w == 1 ? mu_1 : w == 2 ? mu_2 : mu_3);
            wrow = w;
          } else {
            accept = false;
            lam = lam * mu_3;
            keff = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(keff), 48));
            hdmean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hdmean), 48));
            ok = all_small;       // (only read by the exit test of the rejection branch below, together with:)
            finite = all_finite;
            smax = all_small ? 0.f : 1e30f;
          }
        }
        ++my_iters;
        pending = false;
        if (accept) {
          const float rho = (float)((F - Fe) / fmax((double)pred, 1e-30));
          const float tt = 2.f * rho - 1.f;
          float shrink = below_floor ? (1.f / 3.f) : fmaxf(1.f / 3.f, 1.f - tt * tt * tt);
          // (fast decay: the quick way back after a damping jump -- but only for the first few rejections of a solve: a frame
          // that keeps being rejected is cycling between a too small and a too large lambda, and the 1/3 rule damps that)
          if ((bool)((int)(kp.lam_fastdec > 0) & (int)(rho > 0.9f) & (int)(nrej <= kp.fastdec_max_rej))) shrink = kp.lam_fastdec;
          lam = fmaxf(lam * shrink, 1e-9f);
          nu = 2.f;
          F = Fe;
          take = true;
          const bool stalled = (bool)((int)below_floor & (int)(blind >= kp.stall_from) & (int)(smax > kp.stall_ratio * sprev) & (int)(smax < kp.stall_cap * kp.tol));
          blind = below_floor ? blind + 1 : 0;
          sprev = smax;
          const float lam_ok = fmaxf(2.f * delta, 10.f * kp.lam0);  // see dexr_quad.hpp
          if ((bool)(((int)(smax < kp.tol) & ((int)(lam <= lam_ok) | (int)sprint_floor)) | (int)stalled | (int)(blind >= kp.max_blind))) {
            done = true;
            status = ST_CONVERGED;
          } else if (smax < kp.tol) {
            lam = fmaxf(0.1f * lam, 0.5f * lam_ok);
          }
        } else {
          WDIAG(++d_nrej; if (!ok) ++d_nfail;)
          ++nrej;
          lam = fmaxf(lam, 1e-6f) * nu;
          if (kp.lam_jump > 0) lam = fmaxf(lam, kp.lam_jump * (MODCHOL ? hdmean : keff));
          nu *= 2.f;
#pragma unroll
          for (int s = 0; s < NJ2; ++s) xj[s] = xacc[s];
          if (lam > 1e10f) {
            done = true;
            status = finite ? ST_CONVERGED : ST_FALLBACK;
          }
          if ((bool)((int)(ok || MODCHOL) & (int)finite & (int)(smax < kp.tol))) {
            done = true;
            status = ST_CONVERGED;
          }
        }
        done = (bool)((int)done | (int)(my_iters >= kp.max_iter));
      }
      // (a frame that this very pass finished -- the accepted step was below the tolerance: the verification pass of the ladder,
      // which takes no blind steps -- needs no model at its final point: value and kinematics were all the pass had to pay)
      if (SPRINT && __any((bool)((int)take & (int)!done))) {
        // the model at the kept point: its kinematic state and term blocks lie in the frame slot of the row that evaluated it
        const int ws = __builtin_amdgcn_readfirstlane(wrow);
        unsigned char* wb = wbase + L::SLOT0 + (size_t)ws * L::SLOT;
        float *AXo = AXl, *OGo = OGl, *TBo = TBl;
        AXl = reinterpret_cast<float*>(wb + L::AX);
        OGl = reinterpret_cast<float*>(wb + L::OG);
        TBl = reinterpret_cast<float*>(wb + L::TB);
        model_rest();
        AXl = AXo; OGl = OGo; TBl = TBo;
      }
      if (take) {  // the evaluated point becomes the accepted point: keep its gradient and Hessian
#pragma unroll
        for (int s = 0; s < NJ2; ++s) {
          xacc[s] = xj[s];
          if (jin[s]) GVl[jo_[s]] = jopt[s] ? gnew[s] + 2.f * delta * (xj[s] - XLl[jo_[s]]) : 0.f;
        }
        if (!SPRINT) {
#pragma unroll
          for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int j = 0; j <= i / 2; ++j) Ha[i][j] = Hn[i][j];
        }
      }
    }
    // active set of the accepted point, gathered from the 16 lanes of the row
    uint32_t freemask = 0;
    {
      bool fr[NJ2];
#pragma unroll
      for (int s = 0; s < NJ2; ++s) {
        const float2 bx = *reinterpret_cast<const float2*>(BOXw + 2 * (jin[s] ? jo_[s] : 0));
        const float ga = GVl[jin[s] ? jo_[s] : 0];
        const bool act = (bool)(((int)(xacc[s] <= bx.x) & (int)(ga > 0)) | ((int)(xacc[s] >= bx.y) & (int)(ga < 0)));
        fr[s] = (bool)((int)jopt[s] & (int)!act);
      }
      const unsigned long long b0 = __ballot(fr[0]);
      freemask = (uint32_t)((b0 >> (16 * slot)) & 0xFFFFull);
      if (NJ2 > 1) {
        const unsigned long long b1 = __ballot(fr[NJ2 - 1]);
        freemask |= (uint32_t)((b1 >> (16 * slot)) & 0xFFFFull) << 16;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    WPROF_STAGE(5)
    // A failed factorisation (the damped model is indefinite) used to cost a whole pass: the garbage "step" was applied, the
    // kinematics and the assembly ran at that point, and only then did the rejection raise lambda -- 20-33 % of the passes
    // of the frames that need 16+ passes (tools/pass_composition.sh, profiles/r03_wide_pass_composition.txt), i.e. of the tail
    // that bounds a launch.  The rejection needs nothing from that evaluation, so it is taken here: the same lambda / nu /
    // iteration-count updates as the reject branch above, then the factorisation again, inside the pass (16 % of a pass
    // instead of 100 %).  Same sequence of damping values and trial points, hence the same answers.  (One call site in a
    // rolled loop: the unrolled factorisation is the largest block of the kernel.)
    bool okf = true;
#pragma clang loop unroll(disable)
    for (int attempt = 0;; ++attempt) {
      if (SPRINT && !__any((bool)((int)active & (int)!done))) break;  // (the wave's one frame is finished: nothing to step from)
      okf = factor_and_solve(freemask, lam * mu_own);
      // (not at n = 32: the loop around the 32 x 32 factorisation costs 60 more spilled registers, +16 % per launch)
      // (SPRINT: a row whose damped model is indefinite is simply not a candidate; only when NO row has a step is the damping
      // raised here, for all of them, from the largest value tried)
      const bool retry = !MODCHOL && NMAX <= 24 && attempt < 3 && !done && (SPRINT ? __ballot(okf) == 0ull : !okf);  // (uniform over the row's 16 lanes)
      if (!__any(retry)) break;
      if (retry) {
        float gdl = 0.f, ddl = 0.f;
#pragma unroll
        for (int s = 0; s < NJ2; ++s) {  // (selects, not branches: a joint slot that is not free contributes a zero step)
          const int jg = jin[s] ? jo_[s] : 0;
          const float ds = (bool)((int)jin[s] & (int)((freemask >> jg) & 1u)) ? dstep[s] : 0.f;
          gdl -= GVl[jg] * ds;
          ddl += ds * ds;
        }
        const float gd = row_sum(gdl), dd = row_sum(ddl);
        keff = gd / fmaxf(dd, 1e-30f);
        if (SPRINT) {  // (row 3: the largest damping of the ladder)
          keff = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(keff), 48));
          lam = lam * mu_3;
        }
        ++my_iters;
        WDIAG(++d_nrej; ++d_nfail;)
        ++nrej;
        lam = fmaxf(lam, 1e-6f) * nu;
        if (kp.lam_jump > 0) lam = fmaxf(lam, kp.lam_jump * keff);
        nu *= 2.f;
        if (lam > 1e10f) {
          done = true;
          status = ST_CONVERGED;
        }
        if (!done && my_iters >= kp.max_iter) done = true;
      }
    }
    WPROF_STAGE(6)
    const bool stepping = !done;
    if (stepping) {
      ok = okf;
      float dmaxl = 0.f, gdl = 0.f, ddl = 0.f;
      bool son[NJ2];
      int sjg[NJ2];
#pragma unroll
      for (int s = 0; s < NJ2; ++s) {
        sjg[s] = jin[s] ? jo_[s] : 0;
        son[s] = (bool)((int)jin[s] & (int)((freemask >> sjg[s]) & 1u));
        const float ds = son[s] ? dstep[s] : 0.f;
        dmaxl = fmaxf(dmaxl, fabsf(ds));
        gdl -= GVl[sjg[s]] * ds;
        ddl += ds * ds;
      }
      const float dmax = row_max(dmaxl), gd = row_sum(gdl), dd = row_sum(ddl);
      // (MODCHOL: a step from a modified factorisation is stretched towards the trust radius, at most 8 x)
      const float alpha = (bool)((int)(kp.step_cap > 0) & ((int)(dmax > kp.step_cap) | ((int)MODCHOL & (int)!okf & (int)(dmax > 0.f))))
                              ? fminf(kp.step_cap / dmax, MODCHOL ? 8.f : 1e30f) : 1.f;
      WDIAG(if (alpha < 1.f) ++d_ncap;)
      const float lam_row = lam * mu_own;  // (the damping this row's step was computed with)
      pred = alpha * (1.f - 0.5f * alpha) * gd + 0.5f * alpha * alpha * lam_row * dd;
      keff = gd / fmaxf(dd, 1e-30f);
      float sl = 0.f;
#pragma unroll
      for (int s = 0; s < NJ2; ++s) {
        const float2 bx = *reinterpret_cast<const float2*>(BOXw + 2 * sjg[s]);
        const float xt = fminf(fmaxf(xacc[s] + alpha * dstep[s], bx.x), bx.y);
        sl = fmaxf(sl, son[s] ? fabsf(xt - xacc[s]) : 0.f);
        xj[s] = son[s] ? xt : xj[s];
      }
      smax = row_max(sl);
      pending = true;
      // (see dexr_quad.hpp: verified undamped model, tiny Newton step; beyond 10 tol only on the quadratic tail of the
      // iteration -- the step must be at most a tenth of the previous accepted one, as in the small-component kernel)
      bool last_step = (bool)((int)okf & (int)(smax < kp.blind_tol) & (int)(lam_row <= kp.lam0) & ((int)(smax < 10.f * kp.tol) | (int)(smax < 0.1f * sprev)));
      // ... and no variable held at a bound may be about to come off it (round 6).  The step is judged by the model of the FREE
      // variables; a held variable i stays held only while its multiplier g_i keeps its sign, and the step changes it by
      // (H d)_i.  A frame whose held thumb joint had g_i = 1.4e-5 and (H d)_i = -3.7e-5 took a 1.1e-5 rad blind step and ended
      // 1.5e-3 rad from the minimum the verified iteration reaches two passes later (LEAP DexPilot, tools/data/
      // r06_straddle_far_frames.npz frame 1041; the host emulation tools/lm_lab.py reproduces it pass by pass).  |(H d)_i| is
      // bounded by n_free x max diag(H) x |d|_inf (H's Gauss-Newton part is positive semi-definite); a quarter of that bound
      // (8.7 x the failure's own ratio; lab: +0.9 % passes on Shadow DexPilot, none on LEAP, profiles/r06_blind_step_guard.txt)
      // marks a weakly held variable, and such a frame verifies its last step like any other.
      if (__any(last_step)) {
        float hdl = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if ((bool)((int)(a == b) & (int)((freemask >> (4 * i + a)) & 1u))) hdl = fmaxf(hdl, (i & 1) ? Ha[i][i / 2].y : Ha[i][i / 2].x);
        const float tau = 0.25f * (float)__popc(freemask) * (row_max(hdl) + 2.f * delta) * smax;
        bool weakl = false;
#pragma unroll
        for (int s = 0; s < NJ2; ++s)
          weakl = (bool)((int)weakl | ((int)jin[s] & (int)jopt[s] & (int)!son[s] & (int)(fabsf(GVl[sjg[s]]) < tau)));
        last_step = (bool)((int)last_step & (int)!(row_max(weakl ? 1.f : 0.f) > 0.f));
      }
      if (SPRINT) {  // any row's step qualifies: the least damped such row's point is the answer, for every row
        const unsigned long long lb = __ballot(last_step);
        if (lb != 0ull) {
          const int r = (__ffsll((long long)lb) - 1) >> 4;
          const int src = (16 * r + l) << 2;
#pragma unroll
          for (int s2 = 0; s2 < NJ2; ++s2) xj[s2] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(xj[s2])));
          last_step = true;
        }
      }
      if (last_step) {
        ++my_iters;
        pending = false;
        done = true;
        status = ST_CONVERGED;
#pragma unroll
        for (int s = 0; s < NJ2; ++s) xacc[s] = xj[s];
      }
    }

    WPROF_STAGE(7)
    // (last) retire finished frames
    if (active && done) {
      ColdParams& kp = cold();  // (shadows the kernel argument inside this cold path)
      bool badl = false;
#pragma unroll
      for (int s = 0; s < NJ2; ++s)
        if (jopt[s]) badl = badl || !(xacc[s] == xacc[s]);
      const bool bad = row_max(badl ? 1.f : 0.f) > 0.f;
      if (bad) status = ST_FALLBACK;
      const bool writer = !SPRINT || slot == 0;  // (SPRINT: the four rows hold the same answer)
#pragma unroll
      for (int s = 0; s < NJ2; ++s)
        if (jopt[s] && writer) {
          const float v = bad ? XLl[jo_[s]] : xacc[s];
          const int api = tb.api[jsel(s)];
          const int64_t irow = f_irow();
          kp.qout[irow * ld + api] = v;
          if (kp.qout64) kp.qout64[irow * ld + api] = (double)v;
        }
      if (l == 0 && writer) {
        const int64_t irow = f_irow();
        if (kp.status) atomicMax(&kp.status[irow], status);
#ifdef DEXR_WIDE_DIAG
        if (kp.iters) atomicMax(&kp.iters[irow], my_iters | (d_nrej << 8) | (d_ncap << 16) | (d_nfail << 24));
#else
        if (kp.iters) atomicMax(&kp.iters[irow], my_iters);
#endif
        if (kp.fval) atomicAdd(&kp.fval[irow], (float)F);
      }
      const int t_seq = seq ? FS32[0] : 0;
      if (seq && t_seq + 1 < kp.T) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the next frame's start point is the row just written
        load_frame(FS64[2], t_seq + 1);
        load_target();
        reset_state();
      } else {
        if (l == 0 && writer && dexpilot && kp.state && comp == 0) kp.state[f_lrow()] = f_nst();
        active = false;
        done = true;
      }
    }
    WPROF_STAGE(8)
#ifdef DEXR_WIDE_PROF
    wp_acc[11] += 1;
#endif
  }
#ifdef DEXR_WIDE_PROF
  if (kp.g64out && blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 12; ++i) atomicAdd(&kp.g64out[i], (double)wp_acc[i]);
#endif
  if (kp.screen && l == 0 && screen_acc != 0.f) atomicAdd(kp.screen_sum, screen_acc);
}

}  // namespace dexr
