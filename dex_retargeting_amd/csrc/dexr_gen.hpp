// dexr_gen.hpp -- the GENERAL solve kernel: any model the generic table format can describe (include/dexr_tables.h:
// up to 64 joints / variables / target links / reference rows in one component, any tree shape, mimic joints folded,
// 2..8 DexPilot fingers).  It exists so that nothing the reference's optimizers accept
// (/root/reference/src/dex_retargeting/optimizer.py:18-52: any URDF, any number of links and vectors) ends in a
// ValueError here; the specialised families (dexr_kernel / dexr_wide / dexr_red) stay the fast path for everything that
// fits them.
//
// Mapping: ONE WAVEFRONT PER FRAME, float64 throughout, rolled loops over tables of any size; the lanes of the wave talk
// through LDS, which executes one wave's instructions in order -- a write by one lane is seen by every later read of the
// wave, so the "barriers" below are compiler fences (gen_sync), never a wait:
//   joint values           lane k = joint k: everything the tables say about joint / variable / link / term k lives in lane
//                          k's REGISTERS for the whole kernel (placement, axis, box, parent, term masks, targets)
//   forward kinematics     all joints form their local transform X_k . motion(q_k) at once; then level-synchronous over the
//                          tree: the joints of depth d (one lane each) compose their parent's world transform (LDS) with it
//                          -- depth, not joint count, steps of 36 multiply-adds each
//   frames / terms         lane f = target link f; lane t = residual term t (SmoothL1 value, gradient, curvature), which
//                          publishes one 16-double record per term (both frame positions, force, unit vector, curvatures)
//   gradient / Hessian     per term: lane k forms joint k's column a x (p - o) from its own registers + the term's record,
//                          lane v folds its variable's joint family (kinematics_adaptor.py:102-113); the lower triangle of H
//                          is tiled over an 8 x 8 LANE GRID -- lane (a, b) owns the entries (a + 8 i, b + 8 j) -- and
//                          accumulated over the terms in registers: per term a lane reads the <= NI columns of its rows and
//                          the <= NI of its columns (round 4 first half: entry e on lane e mod 64, six reads per entry and
//                          term); the second-order kinematic term is added once per pass from per-joint sums CF_k
//   Cholesky / solves      lane = row in registers; the finished part of the factor is published row-major in LDS, so row c
//                          is read as wave-uniform pairs where Cholesky-Crout needs it and column `lane` where the backward
//                          substitution does; that copy overlays the kinematic state, which is dead by then
// Solver: the projected Levenberg-Marquardt / Newton iteration on F = f + norm_delta |x - last|^2 that
// oracle/solvers.solve_lm_batched states (exact SmoothL1 curvature, second-order kinematic term, Nielsen damping), plus
// the trust radius the other kernels use.  What is computed per evaluation follows the reference's closures
// (optimizer.py:146-198, 249-304, 510-575) and the DexPilot pre-amble (:462-508).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dexr_kernel.hpp"

namespace dexr {

struct GenTab {  // device pointers into the uploaded generic table
  int32_t nj, nf, nt, nv, nfam, max_depth, has_kp;
  const double *X, *axis, *jmul, *joff, *lo, *hi, *frame_off;
  const unsigned long long *frame_anc, *joint_anc;
  const int32_t *jtype, *parent, *depth, *src_idx, *var, *var_api, *fam_off, *fam, *frame_joint, *term_task, *term_origin,
      *term_ref, *row_ho, *row_ht;
};

// LDS of one wave, in doubles.  [scratch | live | tables]: "scratch" is everything an evaluation / model assembly
// produces and the factorisation no longer needs -- world transforms, frame positions, term records, column buffers --
// and the row-major copy of the Cholesky factor (gen_factor_solve) lies over it.
__host__ __device__ inline size_t gen_scratch_doubles(int nj, int nf, int nt, int nv) {
  const size_t kin = (size_t)nj * 12 /* Tw */ + (size_t)nv * 8 /* colA | colB */ + (size_t)nt * 16 /* rec */ +
                     (size_t)nf * 3 /* P */ + (size_t)nj * 3 /* jcol */ + (size_t)nj /* tmp */;
  const size_t ntri = (size_t)nv * (nv + 1) / 2;
  return ((kin > ntri ? kin : ntri) + 1) & ~(size_t)1;
}
__host__ __device__ inline size_t gen_lds_doubles(int nj, int nf, int nt, int nv, int nfam) {
  const size_t live = (size_t)nv * 5 /* x xt g s act */ + (size_t)nv * (nv + 1) / 2 /* H */;
  // wave-local copies of the tables the loops index with data-dependent subscripts (see "tables" in the kernel)
  const size_t ints = 3 * (size_t)nj + (size_t)nv + 1 + (size_t)nfam + (size_t)nf + 2 * (size_t)nt;
  return gen_scratch_doubles(nj, nf, nt, nv) + live + (size_t)nj /* jmul */ + (size_t)nf + (size_t)nj /* masks */ + (ints + 1) / 2;
}

#define GEN_TRI(r, c) ((size_t)(r) * ((r) + 1) / 2 + (c))
// -DDEXR_GEN_PROF=1 (tools/prof_gen_stages.sh; never in the shipped library): block 0 accumulates the core cycles of every
// stage of its passes and adds them to kp.g64out[stage] at the end (solve mode leaves that pointer unused otherwise).
#ifdef DEXR_GEN_PROF
#define GPROF_DECL long long gp_t0 = 0, gp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define GPROF_START() gp_t0 = clock64()
#define GPROF_STAGE(i) { const long long gp_t1 = clock64(); gp_acc[i] += gp_t1 - gp_t0; gp_t0 = gp_t1; }
#define GPROF_COUNT(i) gp_acc[i] += 1
#define GPROF_FLUSH() if (MODE == MODE_SOLVE && blockIdx.x == 0 && lane == 0 && kp.g64out) { for (int i = 0; i < 12; ++i) atomicAdd(&kp.g64out[i], (double)gp_acc[i]); }
#else
#define GPROF_DECL
#define GPROF_START()
#define GPROF_STAGE(i)
#define GPROF_COUNT(i)
#define GPROF_FLUSH()
#endif
// The 8 x 8 lane grid holds NI x NI tiles: NI = 5 serves up to 40 variables (15 entries of the lower triangle per lane, the
// factorisation's 40 rows in registers, two waves per SIMD), NI = 8 up to 64 (36 entries, one wave per SIMD)
constexpr int GEN_NI_TINY = 3, GEN_NI_SMALL = 5, GEN_NI_BIG = 8;  // (TINY, round 6: models of <= 24 variables -- the hands' own configs on
                                                                   // generic tables, the all-float64 path of the sixteen-lane family)

typedef double gen_d2 __attribute__((ext_vector_type(2)));

// Lanes of the wave exchange data through LDS.  A block IS one wave and the LDS executes a wave's instructions in order,
// so a read issued after a write sees it: all that is needed is that the compiler keeps the order (the same idiom as the
// sixteen-lane kernel's, dexr_wide.hpp).  __syncthreads() would add a wait for the writes to drain before the first read
// is even issued -- measured on the first version of this kernel: 19 kinematic levels + ~25 term barriers per pass.
__device__ __forceinline__ void gen_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int CTRL> __device__ __forceinline__ double gen_dpp(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// a double of lane `src` as a wave-uniform value: two v_readlane_b32
__device__ __forceinline__ double gen_readlane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// sum / maximum over the wave, in every lane: four DPP steps inside the rows of 16 lanes, the four row results through
// scalar registers (the first version went through ds_bpermute: six dependent LDS round trips per reduction, five
// reductions per pass)
__device__ __forceinline__ double gen_wave_sum(double v) {
  v += gen_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += gen_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += gen_dpp<0x141>(v);  // row_half_mirror
  v += gen_dpp<0x140>(v);  // row_mirror
  return (gen_readlane(v, 0) + gen_readlane(v, 16)) + (gen_readlane(v, 32) + gen_readlane(v, 48));
}
__device__ __forceinline__ double gen_wave_max(double v) {
  v = fmax(v, gen_dpp<0xB1>(v));
  v = fmax(v, gen_dpp<0x4E>(v));
  v = fmax(v, gen_dpp<0x141>(v));
  v = fmax(v, gen_dpp<0x140>(v));
  return fmax(fmax(gen_readlane(v, 0), gen_readlane(v, 16)), fmax(gen_readlane(v, 32), gen_readlane(v, 48)));
}

// LDS by ADDRESS.  The factorisation below is a real call (not inlined, see there).  Handed generic pointers it reads the
// Hessian with flat loads, and since the only call site passes the kernel's dynamic LDS the compiler specialised the
// pointers into a per-kernel lookup of that block's offset (llvm.amdgcn.dynlds.offset.table): one scalar load + wait in
// every column.  So the function takes 32-bit LDS addresses (made opaque at the call site) and reads / writes through
// address-space-3 pointers: plain ds_read / ds_write with immediate offsets.
typedef __attribute__((address_space(3))) double gen_lds_f64;
typedef __attribute__((address_space(3))) gen_d2 gen_lds_d2;
__device__ __forceinline__ unsigned gen_lds_addr(const double* p) {
  unsigned a = (unsigned)(uintptr_t)(const gen_lds_f64*)p;
  asm volatile("" : "+v"(a));  // (opaque, then wave-uniform again: the idiom of tip_pin, dexr_tip.hpp)
  return (unsigned)__builtin_amdgcn_readfirstlane((int)a);
}

// Solve (H restricted to the free set + lam I) d = -g for the step d.  Lane r holds row r of the lower triangle in
// REGISTERS (NV doubles, statically indexed in fully unrolled loops); the factor is formed column by column
// (Cholesky-Crout): entry (r, c) = (A[r][c] - sum_{k<c} L[r][k] L[c][k]) / L[c][c].  Every finished entry is also written to
// `Lt`, a row-major packed copy of the factor in LDS, so that
//   * row c's entries L[c][k] are read as wave-uniform, contiguous pairs (one ds_read_b128 per two entries; round 4's first
//     version fetched each from lane c with two v_readlane: 3 instructions per multiply-add instead of 1.5) -- the last
//     two entries of the row, written one and two columns ago, still come from lane c, which keeps the LDS write -> read
//     latency off the column-to-column chain;
//   * the backward substitution L^T s = y reads COLUMN `lane` of the factor with one ds_read_b64 per entry (the register
//     variant it replaces carried y as wave-uniform values in every lane and fetched row k from lane k: as many
//     instructions again as the factorisation).
// `Lt` costs no LDS: it overlays the kinematic state of the last evaluation, which nothing reads any more once the model
// has been assembled (gen_scratch_doubles).  The register rows hold the STRICT lower triangle (zeros on and above the
// diagonal, 1 / L[r][r] apart): a substitution step is then the same three instructions in every lane -- broadcast of
// lane k's value, one multiply-add with L[k], no lane == k / lane > k cases -- and the answer is read off at the end.
// Not inlined: inside the kernel the register allocator carried ~400 live registers through the unrolled columns.
// Returns false when a pivot is not positive (the damped model is indefinite: the caller raises lambda).
template <int NV>
__device__ __noinline__ bool gen_factor_solve(int lane, int nv, unsigned aH, unsigned ag, unsigned aact, double lam, unsigned aLt,
                                              unsigned as_out) {
  const gen_lds_f64* H = reinterpret_cast<const gen_lds_f64*>((uintptr_t)aH);
  const gen_lds_f64* g = reinterpret_cast<const gen_lds_f64*>((uintptr_t)ag);
  const gen_lds_f64* act = reinterpret_cast<const gen_lds_f64*>((uintptr_t)aact);
  gen_lds_f64* Lt = reinterpret_cast<gen_lds_f64*>((uintptr_t)aLt);
  gen_lds_f64* s_out = reinterpret_cast<gen_lds_f64*>((uintptr_t)as_out);
  nv = __builtin_amdgcn_readfirstlane(nv);  // (arguments arrive in vector registers: say that this one is wave-uniform)
  const bool rowv = lane < nv;
  const unsigned long long held = __ballot(rowv && act[rowv ? lane : 0] != 0.0);  // bit v: variable v sits at a bound
  const bool a_r = !rowv || ((held >> lane) & 1ull);
  const int base = rowv ? lane * (lane + 1) / 2 : 0;  // (lanes beyond the matrix read row 0 and keep nothing)
  const gen_lds_f64* Hr = H + base;
  double L[NV];
  const double my_diag = a_r ? 1.0 : Hr[lane] + lam;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    double v = 0.0;
    if (8 * (c / 8) < nv) {  // wave-uniform, per group of eight columns
      const bool keep = c < lane && !a_r && !((held >> c) & 1ull);  // (rows / columns of held variables: identity)
      const double h = Hr[c];  // (beyond the lane's row for c > lane: read and dropped)
      v = keep ? h : 0.0;
      v = c == lane ? my_diag : v;
    }
    L[c] = v;
  }
  double rhs = (rowv && !a_r) ? -g[lane] : 0.0;
  gen_lds_f64* my_row = Lt + base;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    if (c < nv) {  // wave-uniform
      double acc = L[c];
      // row c of the factor so far: entries [0, nl) from the LDS copy (wave-uniform, contiguous: pairs where the address is
      // 16-byte aligned).  In the first columns the last two entries -- written one and two columns ago -- come from lane
      // c's registers (the LDS write -> read latency would sit on the column-to-column chain); from column 8 on the reads
      // are issued a column's worth of multiply-adds before their values are due
      const int NEAR = c < 8 ? 2 : 0;
      const int nl = c > NEAR ? c - NEAR : 0;  // (c is the unrolled loop's constant: all of this folds)
      const int rs = c * (c + 1) / 2;
      const gen_lds_f64* row_c = Lt + rs;
      int k = 0;
      if ((rs & 1) && nl > 0) {
        acc = fma(-L[0], row_c[0], acc);
        k = 1;
      }
#pragma unroll
      for (; k + 1 < nl; k += 2) {
        const gen_d2 p = *reinterpret_cast<const gen_lds_d2*>(row_c + k);
        acc = fma(-L[k], p.x, acc);
        acc = fma(-L[k + 1], p.y, acc);
      }
      if (k < nl) acc = fma(-L[k], row_c[k], acc);
#pragma unroll
      for (int k2 = nl; k2 < c; ++k2) acc = fma(-L[k2], gen_readlane(L[k2], c), acc);  // L[k2] of lane c = L[c][k2]
      const double d = gen_readlane(acc, c);
      if (!(d > 0.0)) return false;  // (wave-uniform: d comes out of a scalar register)
      {
        // 1 / sqrt(d): hardware estimate + two Newton steps (full double precision for the well-scaled pivots of a damped
        // Hessian; a correctly rounded sqrt and a division cost three times the instructions, once per column)
        double ip = __builtin_amdgcn_rsq(d);
        ip = ip * fma(-0.5 * d * ip, ip, 1.5);
        ip = ip * fma(-0.5 * d * ip, ip, 1.5);
        const double lv = acc * ip;
        if (rowv && lane >= c) my_row[c] = lv;  // (lane c's own entry is sqrt(d), the diagonal)
        L[c] = lane > c ? lv : 0.0;
        gen_sync();
      }
    }
  }
  // 1 / L[lane][lane] from the diagonal the lane wrote (hardware estimate + two Newton steps)
  double invd = 0.0;
  if (rowv) {
    const double dg = my_row[lane];
    double r = __builtin_amdgcn_rcp(dg);
    r = r * fma(-dg, r, 2.0);
    r = r * fma(-dg, r, 2.0);
    invd = r;
  }
  // L y = rhs, column-wise: once the columns before k have been subtracted, lane k's value is final and is broadcast
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (8 * (k / 8) < nv) {
      const double yk = gen_readlane(rhs * invd, k);
      rhs = fma(-L[k], yk, rhs);  // (rows <= k hold 0 there)
    }
  }
  double y = rhs * invd;
  // L^T s = y the same way: lane c now holds column c (entries L[k][c], k > c), read from the row-major copy
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double v = 0.0;
    if (8 * (k / 8) < nv) {
      const double h = Lt[k * (k + 1) / 2 + lane];
      v = (k > lane && k < nv) ? h : 0.0;
    }
    L[k] = v;
  }
#pragma unroll
  for (int k = NV - 1; k >= 0; --k) {
    if (8 * (k / 8) < nv) {
      const double sk = gen_readlane(y * invd, k);
      y = fma(-L[k], sk, y);
    }
  }
  gen_sync();
  if (rowv) s_out[lane] = y * invd;
  return true;
}

template <int MODE, int NI>
#ifndef DEXR_GEN_SMALL_MINW
#define DEXR_GEN_SMALL_MINW 2
#endif
__global__ void __launch_bounds__(64, NI <= GEN_NI_TINY ? 2 : (NI == GEN_NI_SMALL ? DEXR_GEN_SMALL_MINW : 1)) dexr_gen_kernel(KernelParams kp, GenTab tb) {
  extern __shared__ __align__(16) double gen_lds[];
  constexpr int NSLOT = NI * (NI + 1) / 2;  // entries of the lower triangle a lane of the grid owns: tiles (i, j), j <= i
  constexpr int NV = NI * 8;                // rows the register factorisation holds
  const int lane = threadIdx.x;
  const int nj = tb.nj, nf = tb.nf, nt = tb.nt, nv = tb.nv;
  // ---- scratch (dead once the model of a point has been assembled; the factor's copy lies over it) ------------------------
  double* Tw = gen_lds;             // nj x 12: world transform of every joint frame AFTER its motion (R row-major | p)
  double* colA = Tw + nj * 12;      // nv x 4: a term's column per variable | u . column   (16-byte aligned records)
  double* colB = colA + nv * 4;     // nv x 4: the second term of a pair
  double* rec = colB + nv * 4;      // nt x 16: per term  task pos (3) | origin pos (3) | force (3) | unit vector (3) | curvature (3 + 1)
  double* P = rec + nt * 16;        // nf x 3
  double* jcol = P + nf * 3;        // nj x 3
  double* tmp = jcol + nj * 3;      // nj
  int32_t* jptr = reinterpret_cast<int32_t*>(tmp);  // nj: the kinematics' ancestor pointers (tmp is the model assembly's)
  double* Lt = gen_lds;             // row-major packed Cholesky factor (gen_factor_solve)
  // ---- live --------------------------------------------------------------------------------------------------------------
  double* x = gen_lds + gen_scratch_doubles(nj, nf, nt, nv);  // nv each:
  double* xt = x + nv;
  double* g = xt + nv;
  double* s = g + nv;
  double* act = s + nv;
  const int ntri = nv * (nv + 1) / 2;
  double* H = act + nv;            // lower triangle, packed by rows: entry (r, c), c <= r, at r (r + 1) / 2 + c
  // tables the loops index with data-dependent subscripts (families of a variable, ancestor masks): read from global memory
  // they cost two dependent ~1 us round trips per access, so every wave keeps its own copy in LDS
  double* l_jmul = H + ntri;                                                        // nj
  unsigned long long* l_fanc = reinterpret_cast<unsigned long long*>(l_jmul + nj);  // nf
  unsigned long long* l_janc = l_fanc + nf;                                       // nj
  int32_t* l_jtype = reinterpret_cast<int32_t*>(l_janc + nj);                     // nj
  int32_t* l_parent = l_jtype + nj;                                               // nj
  int32_t* l_var = l_parent + nj;                                                 // nj
  int32_t* l_famoff = l_var + nj;                                                 // nv + 1
  int32_t* l_fam = l_famoff + nv + 1;                                             // nfam
  int32_t* l_fjoint = l_fam + tb.nfam;                                            // nf
  int32_t* l_ttask = l_fjoint + nf;                                               // nt
  int32_t* l_torigin = l_ttask + nt;                                              // nt
  for (int i = lane; i < nj; i += 64) {
    l_jmul[i] = tb.jmul[i];
    l_janc[i] = tb.joint_anc[i];
    l_jtype[i] = tb.jtype[i];
    l_parent[i] = tb.parent[i];
    l_var[i] = tb.var[i];
  }
  for (int i = lane; i < nf; i += 64) {
    l_fanc[i] = tb.frame_anc[i];
    l_fjoint[i] = tb.frame_joint[i];
  }
  for (int i = lane; i <= nv; i += 64) l_famoff[i] = tb.fam_off[i];
  for (int i = lane; i < tb.nfam; i += 64) l_fam[i] = tb.fam[i];
  for (int i = lane; i < nt; i += 64) {
    l_ttask[i] = tb.term_task[i];
    l_torigin[i] = tb.term_origin[i];
  }
  __syncthreads();

  // Lane k IS joint k (and variable k, link k, term k) for the whole kernel: what the tables say about them is read ONCE,
  // into registers.
  const bool is_j = lane < nj, is_v = lane < nv, is_f = lane < nf, is_t = lane < nt;
  const int my_src = is_j ? tb.src_idx[lane] : 0;
  const int my_var = is_j ? l_var[lane] : -1;
  const int my_parent = is_j ? l_parent[lane] : -1;
  const bool my_rev = is_j && l_jtype[lane] == DEXR_JOINT_REVOLUTE;
#define my_jmul (is_j ? l_jmul[lane] : 0.0)  /* the wave's LDS copy, at use (registers are the scarce resource) */
  // (the placement, local axis and frame offset of the lane's joint / link -- 21 doubles -- are read again from the tables at
  // the top of every evaluation: resident in L2, issued before the sines and cosines they wait behind; held in registers
  // across the whole pass they were 42 of the VGPRs a kernel at 256 spills)
  const int my_fjoint = is_f ? l_fjoint[lane] : -1;
#define my_lo (is_v ? tb.lo[lane] : 0.0)  /* box of the lane's variable: read from the tables where the step is clipped */
#define my_hi (is_v ? tb.hi[lane] : 0.0)
  const int my_api = is_v ? tb.var_api[lane] : 0;
  const int my_ft = is_t ? l_ttask[lane] : 0, my_fo_t = is_t ? l_torigin[lane] : -1;  // lane t: frames of term t
  // terms whose task / origin chain joint `lane` lies on (bit t): the column of joint k for term t needs nothing else
  unsigned long long on_task = 0ull, on_origin = 0ull;
  for (int t = 0; t < nt; ++t) {
    const int ft = l_ttask[t], fo = l_torigin[t];
    if (is_j && ((l_fanc[ft] >> lane) & 1ull)) on_task |= 1ull << t;
    if (is_j && fo >= 0 && ((l_fanc[fo] >> lane) & 1ull)) on_origin |= 1ull << t;
  }
  const bool any_prismatic = __any(is_j && !my_rev);  // wave-uniform
  const int kin_rounds = tb.max_depth > 0 ? 32 - __builtin_clz((unsigned)tb.max_depth) : 0;  // 2^rounds - 1 >= depth of the deepest joint
  const bool no_mimic = tb.nfam == nv;  // every variable drives exactly one joint (families partition the driven joints)

  // The lane's tiles of the lower triangle: lane (a, b) of the 8 x 8 grid owns entry (a + 8 i, b + 8 j) for j <= i (on the
  // diagonal tiles only where b <= a); slot i (i + 1) / 2 + j.
  const int ga = lane >> 3, gb = lane & 7;
  int row_ld[NI], col_ld[NI];  // rows / columns to LOAD (clamped into the matrix: an out-of-range tile computes garbage it never stores)
  int row_tri[NI];             // start of row ga + 8 i in the packed triangle
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    row_ld[i] = min(ga + 8 * i, nv - 1);
    col_ld[i] = min(gb + 8 * i, nv - 1);
    row_tri[i] = (ga + 8 * i) * (ga + 8 * i + 1) / 2;
  }
  unsigned long long slot_ok = 0ull, so_rc = 0ull, so_cr = 0ull;  // per slot: entry exists | second-order contributions (mimic-free models)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const int r = ga + 8 * i, c = gb + 8 * j, sl = i * (i + 1) / 2 + j;
      if (r < nv && c <= r) {
        slot_ok |= 1ull << sl;
        if (no_mimic) {
          const int jr = l_fam[l_famoff[r]], jc = l_fam[l_famoff[c]];
          // joint of r moved, joint of c its revolute ancestor-or-self; and the other way round
          if (((l_janc[jr] >> jc) & 1ull) && l_jtype[jc] == DEXR_JOINT_REVOLUTE) so_rc |= 1ull << sl;
          if (jr != jc && ((l_janc[jc] >> jr) & 1ull) && l_jtype[jr] == DEXR_JOINT_REVOLUTE) so_cr |= 1ull << sl;
        }
      }
    }
  }

  GPROF_DECL
  const int64_t cnt = kp.bucket ? (int64_t)kp.bucket[1] : kp.B;
  const int64_t boff = kp.bucket ? (int64_t)kp.bucket[0] : 0;
  const int ld = kp.ld;
  const bool seq = kp.T > 0;
  const int n_frames = seq ? kp.T : 1;
  const bool dexpilot = kp.kind == DEXR_KIND_DEXPILOT;
  const bool pos = kp.kind == DEXR_KIND_POSITION;
  const double beta = (double)kp.huber_delta, delta = (double)kp.norm_delta, inv_norm = (double)kp.inv_norm;
  const double ibeta = 1.0 / beta;
  double my_o[3] = {0, 0, 0}, my_a[3] = {0, 0, 0};  // lane k: origin and world axis of joint k at the last evaluated point

  for (int64_t item = blockIdx.x; item < cnt; item += gridDim.x) {
    const int64_t r0 = kp.perm ? (int64_t)kp.perm[boff + item] : item;
    uint32_t st_carry = 0u;
    double my_xl = 0.0;  // lane v: regularisation target of variable v
    for (int t_seq = 0; t_seq < n_frames; ++t_seq) {
      const int64_t it = seq ? (int64_t)t_seq * kp.seq_stride + r0 : r0;  // row of this frame's inputs / outputs
      const bool carry = seq && t_seq > 0;
      auto ref_row = [&](int row, float* rv) {
        if (kp.kpts) {
          const int ho = tb.row_ho[row], ht = tb.row_ht[row];
          const float* k = kp.kpts + it * (int64_t)kp.n_kp * 3;
#pragma unroll
          for (int i = 0; i < 3; ++i) rv[i] = ho >= 0 ? k[ht * 3 + i] - k[ho * 3 + i] : k[ht * 3 + i];
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) rv[i] = kp.ref[(it * kp.n_ref + row) * 3 + i];
        }
      };
      // ---- load the frame -------------------------------------------------------------------------------------------
      if (MODE != MODE_FK && lane < nv) {
        const int api = my_api;
        double v, l;
        if (MODE == MODE_EVAL) v = kp.xin[it * kp.n_opt + api];
        else if (carry) v = (double)(float)x[lane];  // the reference carries the float32 result (optimizer.py:99)
        else if (kp.x0) v = (double)kp.x0[r0 * ld + api];
        else v = (double)kp.last[r0 * ld + api];
        l = carry ? v : (double)kp.last[r0 * ld + api];
        if (seq) {  // seq_retarget.py:118-120: last_qpos clipped to the joint limits before every solve
          l = fmin(fmax(l, my_lo + (double)kp.clip_eps), my_hi - (double)kp.clip_eps);
          v = l;
        }
        x[lane] = v;
        my_xl = l;
      }
      // joints that follow a caller-fixed value: constant over the frame's passes
      double my_qfix = 0.0;
      if (MODE == MODE_FK) {
        if (is_j) my_qfix = kp.xin[it * kp.n_q + my_src];
      } else if (is_j && my_var < 0) {
        my_qfix = my_jmul * (double)kp.fixed[it * kp.ldf + my_src] + tb.joff[lane];
      }
      uint32_t nst = 0;
      double my_tgt[3] = {0, 0, 0}, my_wt = 1.0;  // lane t: target and weight of term t
      if (MODE != MODE_FK) {
        if (dexpilot) {  // optimizer.py:462-508 (every lane redundantly: wave-uniform)
          const int F = kp.num_fingers;
          const int n_pair = F * (F - 1) / 2, len_s1 = F - 1;
          const uint32_t st = carry ? st_carry : (kp.state ? kp.state[r0] : 0u);
          for (int i = 0; i < len_s1; ++i) {
            float rv[3];
            ref_row(i, rv);
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
              ref_row(idx, rv);
              const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
              const bool b = ((nst >> b2) & 1u) && ((nst >> a) & 1u) && (dist <= 0.03f);
              nst |= (b ? 1u : 0u) << idx;
              ++idx;
            }
          if (is_t) {
            const int row = tb.term_ref[lane];
            float rv[3], tv[3], w;
            ref_row(row, rv);
            if (row < n_pair) {
              if ((nst >> row) & 1u) {
                const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
                const float eta = row < len_s1 ? kp.eta1 : kp.eta2;
                for (int i = 0; i < 3; ++i) tv[i] = (rv[i] / (dist + 1e-6f)) * eta;
                w = row < len_s1 ? 200.f : 400.f;
              } else {
                for (int i = 0; i < 3; ++i) tv[i] = rv[i] * kp.scaling;
                w = 1.f;
              }
            } else {
              for (int i = 0; i < 3; ++i) tv[i] = rv[i] * kp.scaling;
              w = (float)(n_pair + F);
            }
            for (int i = 0; i < 3; ++i) my_tgt[i] = (double)tv[i];
            my_wt = (double)w;
          }
        } else if (is_t) {
          const float sc = (kp.kind == DEXR_KIND_VECTOR) ? kp.scaling : 1.f;
          float rv[3];
          ref_row(tb.term_ref[lane], rv);
          for (int i = 0; i < 3; ++i) my_tgt[i] = (double)(rv[i] * sc);  // f32 multiply: optimizer.py:246
        }
      }
      gen_sync();

      // ---- one evaluation at xs: kinematics, terms, value (eval_value); gradient + Hessian of the data term at the point
      // whose kinematic state is in LDS / the lanes' registers (assemble_model) ----------------------------------------------
      auto eval_value = [&](const double* xs) -> double {
        GPROF_START();
        // every joint's local transform X_k . motion(q_k) at once (sines / cosines of all joints in one go, not one level at
        // a time): rotation about / translation along the joint's own axis
        double Lc[12];
        double my_ax[3], my_fo[3];
        {
          int lane_o = lane;
          asm volatile("" : "+v"(lane_o));  // (opaque: the loads below stay inside the evaluation)
          double myX[12];
          const int jl = is_j ? lane_o : 0, fl = is_f ? lane_o : 0;
#pragma unroll
          for (int i = 0; i < 12; ++i) myX[i] = tb.X[(size_t)jl * 12 + i];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            my_ax[i] = tb.axis[jl * 3 + i];
            my_fo[i] = tb.frame_off[(size_t)fl * 3 + i];
          }
          const double my_joff = tb.joff[jl];
          if (!is_j) {  // (lanes beyond the joints used to carry zeros)
#pragma unroll
            for (int i = 0; i < 12; ++i) myX[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) my_ax[i] = 0.0;
          }
          if (!is_f) {
#pragma unroll
            for (int i = 0; i < 3; ++i) my_fo[i] = 0.0;
          }
          double q = my_qfix;
          if (MODE != MODE_FK && my_var >= 0) q = my_jmul * xs[my_var] + my_joff;
          const double ax = my_ax[0], ay = my_ax[1], az = my_ax[2];
          double sn, cs;
          sincos_f64(my_rev ? q : 0.0, &sn, &cs);  // (dexr_math.hpp: no slow path, unlike ocml's)
          const double c1 = 1.0 - cs;
          // Rodrigues about the local axis: I + sin K + (1 - cos) K^2 (the identity for a translation joint)
          const double M[9] = {1 - c1 * (ay * ay + az * az), -sn * az + c1 * ax * ay, sn * ay + c1 * ax * az,
                               sn * az + c1 * ax * ay, 1 - c1 * (ax * ax + az * az), -sn * ax + c1 * ay * az,
                               -sn * ay + c1 * ax * az, sn * ax + c1 * ay * az, 1 - c1 * (ax * ax + ay * ay)};
          const double tq = my_rev ? 0.0 : q;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) Lc[3 * i + j] = myX[3 * i] * M[j] + myX[3 * i + 1] * M[3 + j] + myX[3 * i + 2] * M[6 + j];
            Lc[9 + i] = myX[9 + i] + tq * (myX[3 * i] * ax + myX[3 * i + 1] * ay + myX[3 * i + 2] * az);
          }
        }
        // World transforms by POINTER JUMPING: every joint holds the product of the local transforms from below its current
        // "pointer" ancestor down to itself; a round composes it with that ancestor's product and adopts the ancestor's
        // pointer, which doubles the length of the covered path -- ceil(log2(depth + 1)) rounds with every lane busy,
        // instead of one round per tree level with one or two lanes busy (19 levels for an arm + hand: a third of the
        // kinematics' instructions and a quarter of its exposed LDS round trips).  A round's reads all precede its writes
        // in the wave's instruction stream, so the old values are what is read.
        double Tk[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Tk[i] = Lc[i];
        {
          int pj = my_parent;
          gen_d2* To = reinterpret_cast<gen_d2*>(Tw + lane * 12);
          if (is_j) {
#pragma unroll
            for (int i = 0; i < 6; ++i) To[i] = gen_d2{Tk[2 * i], Tk[2 * i + 1]};
            jptr[lane] = pj;
          }
          gen_sync();
          for (int r = 0; r < kin_rounds; ++r) {
            const bool go = is_j && pj >= 0;
            if (go) {
              const gen_d2* Tp = reinterpret_cast<const gen_d2*>(Tw + pj * 12);
              const gen_d2 p0 = Tp[0], p1 = Tp[1], p2 = Tp[2], p3 = Tp[3], p4 = Tp[4], p5 = Tp[5];
              pj = jptr[pj];
              const double Rp[9] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y, p4.x};
              const double pp[3] = {p4.y, p5.x, p5.y};
              double Tn[12];
#pragma unroll
              for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) Tn[3 * i + j] = Rp[3 * i] * Tk[j] + Rp[3 * i + 1] * Tk[3 + j] + Rp[3 * i + 2] * Tk[6 + j];
                Tn[9 + i] = Rp[3 * i] * Tk[9] + Rp[3 * i + 1] * Tk[10] + Rp[3 * i + 2] * Tk[11] + pp[i];
              }
#pragma unroll
              for (int i = 0; i < 12; ++i) Tk[i] = Tn[i];
            }
            gen_sync();
            if (go) {
#pragma unroll
              for (int i = 0; i < 6; ++i) To[i] = gen_d2{Tk[2 * i], Tk[2 * i + 1]};
              jptr[lane] = pj;
            }
            gen_sync();
          }
        }
        // world axis (R . M leaves the axis where R put it) and origin of the joint frame: the lane's own registers
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          my_a[i] = Tk[3 * i] * my_ax[0] + Tk[3 * i + 1] * my_ax[1] + Tk[3 * i + 2] * my_ax[2];
          my_o[i] = Tk[9 + i];
        }
        if (is_f) {
          if (my_fjoint < 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) P[lane * 3 + i] = my_fo[i];
          } else {
            const double* T = Tw + my_fjoint * 12;
#pragma unroll
            for (int i = 0; i < 3; ++i) P[lane * 3 + i] = T[3 * i] * my_fo[0] + T[3 * i + 1] * my_fo[1] + T[3 * i + 2] * my_fo[2] + T[9 + i];
          }
        }
        gen_sync();
        GPROF_STAGE(0)  // forward kinematics + frames
        if (MODE == MODE_FK) return 0.0;
        double fpart = 0.0;
        if (is_t) {
          double pt[3], po[3], r[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            pt[i] = P[my_ft * 3 + i];
            po[i] = my_fo_t >= 0 ? P[my_fo_t * 3 + i] : 0.0;
            r[i] = pt[i] - po[i] - my_tgt[i];
          }
          double tgv[3], tuv[3] = {0, 0, 0}, kk[3], kb = 0.0;
          if (pos) {  // SmoothL1 per coordinate, mean over 3 P entries (optimizer.py:163-166)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const double ad = fabs(r[i]);
              const bool in = ad < beta;
              fpart += (in ? 0.5 * r[i] * r[i] * ibeta : ad - 0.5 * beta) * inv_norm;
              tgv[i] = (in ? r[i] * ibeta : (r[i] > 0 ? 1.0 : (r[i] < 0 ? -1.0 : 0.0))) * inv_norm;
              kk[i] = in ? inv_norm * ibeta : 0.0;
            }
          } else {  // SmoothL1 of the vector norm, weighted, mean over the V vectors (optimizer.py:262-273, 523-546)
            // (norm, its reciprocal and the solver's divisions from v_rsq_f64 / v_rcp_f64 + two Newton steps: RealTraits<double>)
            const double d2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
            const double id = d2 > 0 ? RealTraits<double, true>::rsqrt(d2) : 0.0;
            const double d = d2 * id;
            const bool in = d < beta;
            const double w = my_wt;
            fpart = w * (in ? 0.5 * d * d * ibeta : d - 0.5 * beta) * inv_norm;
            const double psi = in ? d * ibeta : 1.0, kq = in ? ibeta : 0.0;
            const double gc = w * psi * inv_norm * id;  // torch.norm backward: zero at 0
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              tgv[i] = gc * r[i];
              tuv[i] = r[i] * id;
              kk[i] = gc;                // H_vec = c1 (I - u u^T) + c2 u u^T
            }
            kb = w * kq * inv_norm - gc;  // (c2 - c1)
          }
          gen_d2* rc = reinterpret_cast<gen_d2*>(rec + lane * 16);
          rc[0] = gen_d2{pt[0], pt[1]};
          rc[1] = gen_d2{pt[2], po[0]};
          rc[2] = gen_d2{po[1], po[2]};
          rc[3] = gen_d2{tgv[0], tgv[1]};
          rc[4] = gen_d2{tgv[2], tuv[0]};
          rc[5] = gen_d2{tuv[1], tuv[2]};
          rc[6] = gen_d2{kk[0], kk[1]};
          rc[7] = gen_d2{kk[2], kb};
        }
        const double fval = gen_wave_sum(fpart);
        gen_sync();
        GPROF_STAGE(1)  // terms
        return fval;
      };
      auto assemble_model = [&]() {
        GPROF_START();
        double hacc[NSLOT];  // this lane's entries of H, accumulated over the terms
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) hacc[i] = 0.0;
        double cf[3] = {0.0, 0.0, 0.0};  // lane k: CF_k (see below)
        double gacc = 0.0;               // gradient entry of the variable this lane writes columns for
        // joint k's column of term t: a x (p - o) over the term's two frames (sign), or the axis (prismatic); p from the
        // term's record, a / o / the chain masks from the lane's registers
        auto joint_column = [&](int t, const double* rc, double (&c)[3]) {
          const gen_d2* r2 = reinterpret_cast<const gen_d2*>(rc);
          const gen_d2 q0 = r2[0], q1 = r2[1], q2 = r2[2], q3 = r2[3], q4 = r2[4];
          const double pt[3] = {q0.x, q0.y, q1.x}, po[3] = {q1.y, q2.x, q2.y}, tgv[3] = {q3.x, q3.y, q4.x};
          // on the task / origin chain of term t?  as 0.0 / 1.0 (t is wave-uniform: the half of the mask is picked with scalar code)
          const unsigned wt_ = t < 32 ? (unsigned)on_task : (unsigned)(on_task >> 32), wo_ = t < 32 ? (unsigned)on_origin : (unsigned)(on_origin >> 32);
          const double sT = (double)((wt_ >> (t & 31)) & 1u), sO = (double)((wo_ >> (t & 31)) & 1u);
          double dv[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) dv[i] = sT * (pt[i] - my_o[i]) - sO * (po[i] - my_o[i]);
          c[0] = my_a[1] * dv[2] - my_a[2] * dv[1];
          c[1] = my_a[2] * dv[0] - my_a[0] * dv[2];
          c[2] = my_a[0] * dv[1] - my_a[1] * dv[0];
          if (any_prismatic) {  // wave-uniform
            const double sg = sT - sO;
#pragma unroll
            for (int i = 0; i < 3; ++i) c[i] = my_rev ? c[i] : sg * my_a[i];
          }
          // second-order kinematic term, first half: CF_k = sum over the terms of (column of joint k) x (force of the
          // term) -- the entry for a pair (joint k, revolute ancestor-or-self j) is m_j m_k a_j . CF_k, formed ONCE per
          // pass after the term loop (tg . (a_j x c) = a_j . (c x tg))
          cf[0] += c[1] * tgv[2] - c[2] * tgv[1];
          cf[1] += c[2] * tgv[0] - c[0] * tgv[2];
          cf[2] += c[0] * tgv[1] - c[1] * tgv[0];
        };
        // Everything that touches the lane's tiles is instantiated for the number NA of tile rows the model has (ceil(nv / 8),
        // rounded up to an instantiated value: tiles past the matrix load clamped rows and are never stored): no per-tile
        // "is this row there" branches inside the term loop.
        auto model = [&](auto POS, auto NA_) {
          constexpr int NA = decltype(NA_)::value;
          // The lane's tile bookkeeping is re-derived HERE from values the compiler cannot see through: left visible it
          // hoists every per-slot address and "does this entry exist" mask out of the frame loop -- 45 values that do not
          // fit, i.e. scratch reloads and v_readlane pairs on the write-out path of every assembly (measured: the stage
          // doubled its time once eight waves shared a CU).
          int gb_o = gb;
          unsigned ok_lo = (unsigned)slot_ok, ok_hi = (unsigned)(slot_ok >> 32), rc_lo = (unsigned)so_rc, rc_hi = (unsigned)(so_rc >> 32),
                   cr_lo = (unsigned)so_cr, cr_hi = (unsigned)(so_cr >> 32);
          asm volatile("" : "+v"(gb_o), "+v"(ok_lo), "+v"(ok_hi), "+v"(rc_lo), "+v"(rc_hi), "+v"(cr_lo), "+v"(cr_hi));
          const int dump_idx = min(lane, nj * 12 - 1);
          auto slot_bit = [](unsigned lo, unsigned hi, int sl) -> unsigned { return ((sl < 32 ? lo : hi) >> (sl & 31)) & 1u; };
          // Entries of H owned by this lane += the contributions of one term: columns (and u . column) of the variables in `cb`
          // (nv x 4), curvatures in the term's record.  position: sum_i k_i v_r[i] v_c[i]; vector kinds:
          // c1 (v_r . v_c) + (c2 - c1) (u . v_r) (u . v_c).  The rows' columns are loaded once, the columns' stream through.
          auto add_entries = [&](const double* cb, const double* rc) {
            const gen_d2* cb2 = reinterpret_cast<const gen_d2*>(cb);
            const gen_d2 k01 = *reinterpret_cast<const gen_d2*>(rc + 12), k2b = *reinterpret_cast<const gen_d2*>(rc + 14);
            gen_d2 Ra[NA], Rb[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
              Ra[i] = cb2[2 * row_ld[i]];
              Rb[i] = cb2[2 * row_ld[i] + 1];
            }
#pragma unroll
            for (int j = 0; j < NA; ++j) {
              const gen_d2 ca = cb2[2 * col_ld[j]], cbv = cb2[2 * col_ld[j] + 1];
              const double w0 = k01.x * ca.x, w1 = k01.y * ca.y, w2 = k2b.x * cbv.x, w3 = k2b.y * cbv.y;
#pragma unroll
              for (int i = j; i < NA; ++i) {
                double h = hacc[i * (i + 1) / 2 + j];
                h = fma(Ra[i].x, w0, h);
                h = fma(Ra[i].y, w1, h);
                h = fma(Rb[i].x, w2, h);
                if (!POS.value) h = fma(Rb[i].y, w3, h);
                hacc[i * (i + 1) / 2 + j] = h;
              }
            }
          };
          if (no_mimic) {
            // every variable drives exactly one joint: lane k writes its variable's column record itself.  Terms in PAIRS:
            // both terms' columns are published (colA, colB) before either is read
            auto own_column = [&](int t, double* cb) {
              const double* rc = rec + t * 16;
              double c[3];
              joint_column(t, rc, c);
              if (my_var >= 0) {
                const double c0 = my_jmul * c[0], c1 = my_jmul * c[1], c2 = my_jmul * c[2];
                const gen_d2 q3 = *reinterpret_cast<const gen_d2*>(rc + 6), q4 = *reinterpret_cast<const gen_d2*>(rc + 8),
                             q5 = *reinterpret_cast<const gen_d2*>(rc + 10);
                gacc += q3.x * c0 + q3.y * c1 + q4.x * c2;
                const double u = POS.value ? 0.0 : q4.y * c0 + q5.x * c1 + q5.y * c2;
                gen_d2* o = reinterpret_cast<gen_d2*>(cb + my_var * 4);
                o[0] = gen_d2{c0, c1};
                o[1] = gen_d2{c2, u};
              }
            };
            for (int t = 0; t < nt; t += 2) {
              const bool two = t + 1 < nt;  // wave-uniform
              if (is_j) {
                own_column(t, colA);
                if (two) own_column(t + 1, colB);
              }
              gen_sync();
              add_entries(colA, rec + t * 16);
              if (two) add_entries(colB, rec + (t + 1) * 16);
              gen_sync();
            }
          } else {
            for (int t = 0; t < nt; ++t) {
              const double* rc = rec + t * 16;
              if (is_j) {
                double c[3];
                joint_column(t, rc, c);
#pragma unroll
                for (int i = 0; i < 3; ++i) jcol[lane * 3 + i] = c[i];
              }
              gen_sync();
              if (is_v) {  // the variable's column: its joint family folded (kinematics_adaptor.py:102-113)
                double c[3] = {0, 0, 0};
                for (int e = l_famoff[lane]; e < l_famoff[lane + 1]; ++e) {
                  const int k = l_fam[e];
                  const double mk = l_jmul[k];
                  for (int i = 0; i < 3; ++i) c[i] += mk * jcol[k * 3 + i];
                }
                gacc += rc[6] * c[0] + rc[7] * c[1] + rc[8] * c[2];
                // u . column (vector kinds only: the position objective has no unit vector)
                const double u = POS.value ? 0.0 : rc[9] * c[0] + rc[10] * c[1] + rc[11] * c[2];
                gen_d2* o = reinterpret_cast<gen_d2*>(colA + lane * 4);
                o[0] = gen_d2{c[0], c[1]};
                o[1] = gen_d2{c[2], u};
              }
              gen_sync();
              add_entries(colA, rc);
              gen_sync();
            }
          }
          // the gradient of the data term: one writer per variable
          if (no_mimic) {
            if (my_var >= 0) g[my_var] = gacc;
          } else if (is_v) {
            g[lane] = gacc;
          }
          GPROF_STAGE(2)  // term loop: columns, gradient, Hessian entries
          // second-order kinematic term, second half: for every joint k that moves with a variable and every revolute
          // ancestor-or-self j of k that does too,  H[var j][var k] += m_j m_k a_j . CF_k  (twice for j != k inside one
          // family: both orders of the unordered pair).
          if (kp.newton && no_mimic) {
            // one joint per variable: lane k publishes m a_k and m CF_k under its variable, the owner of entry (r, c) forms
            // its (at most two) contributions -- the joint of r moved with the joint of c as its ancestor, and the other way
            // round (which of the two exist was worked out per slot when the kernel started)
            if (my_var >= 0) {
              gen_d2* oa = reinterpret_cast<gen_d2*>(colA + my_var * 4);
              gen_d2* oc = reinterpret_cast<gen_d2*>(colB + my_var * 4);
              oa[0] = gen_d2{my_jmul * my_a[0], my_jmul * my_a[1]};
              oa[1] = gen_d2{my_jmul * my_a[2], 0.0};
              oc[0] = gen_d2{my_jmul * cf[0], my_jmul * cf[1]};
              oc[1] = gen_d2{my_jmul * cf[2], 0.0};
            }
            gen_sync();
            const gen_d2* A2 = reinterpret_cast<const gen_d2*>(colA);
            const gen_d2* C2 = reinterpret_cast<const gen_d2*>(colB);
            gen_d2 Ara[NA], Arb[NA], Cra[NA], Crb[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
              Ara[i] = A2[2 * row_ld[i]]; Arb[i] = A2[2 * row_ld[i] + 1];
              Cra[i] = C2[2 * row_ld[i]]; Crb[i] = C2[2 * row_ld[i] + 1];
            }
#pragma unroll
            for (int j = 0; j < NA; ++j) {
              const gen_d2 Aca = A2[2 * col_ld[j]], Acb = A2[2 * col_ld[j] + 1], Cca = C2[2 * col_ld[j]], Ccb = C2[2 * col_ld[j] + 1];
#pragma unroll
              for (int i = j; i < NA; ++i) {
                const int sl = i * (i + 1) / 2 + j;
                const double v1 = Aca.x * Cra[i].x + Aca.y * Cra[i].y + Acb.x * Crb[i].x;  // a_c . CF_r
                const double v2 = Ara[i].x * Cca.x + Ara[i].y * Cca.y + Arb[i].x * Ccb.x;  // a_r . CF_c
                hacc[sl] = fma((double)slot_bit(rc_lo, rc_hi, sl), v1, fma((double)slot_bit(cr_lo, cr_hi, sl), v2, hacc[sl]));
              }
            }
          }
#pragma unroll
          for (int i = 0; i < NA; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
              const int sl = i * (i + 1) / 2 + j;
              // (entries that do not exist are dropped into the dead world transforms: an address select, no branch)
              double* dst = slot_bit(ok_lo, ok_hi, sl) ? H + row_tri[i] + gb_o + 8 * j : Tw + dump_idx;
              *dst = hacc[sl];
            }
          }
          gen_sync();
        };
        auto with_rows = [&](auto POS) {
          if constexpr (NI == GEN_NI_TINY) {
            if (nv <= 16) model(POS, std::integral_constant<int, 2>{});
            else model(POS, std::integral_constant<int, 3>{});
          } else if constexpr (NI == GEN_NI_SMALL) {
            if (nv <= 16) model(POS, std::integral_constant<int, 2>{});
            else if (nv <= 32) model(POS, std::integral_constant<int, 4>{});
            else model(POS, std::integral_constant<int, 5>{});
          } else {
            if (nv <= 48) model(POS, std::integral_constant<int, 6>{});
            else if (nv <= 56) model(POS, std::integral_constant<int, 7>{});
            else model(POS, std::integral_constant<int, 8>{});
          }
        };
        if (pos) with_rows(std::true_type{});
        else with_rows(std::false_type{});
        if (kp.newton && !no_mimic) {
          // families of several joints: one sweep over the joints per PASS -- lane j forms the entry of pair (j, k), lane v
          // sums its variable's family and adds into its own entries.  (Round 3 walked the chains of every TERM instead --
          // terms x chain depth x 2 block barriers, ~1 500 per pass for an arm + hand: most of the ~300 us a pass took.)
          if (is_j) {
#pragma unroll
            for (int i = 0; i < 3; ++i) jcol[lane * 3 + i] = cf[i];
          }
          gen_sync();
          for (int k = 0; k < nj; ++k) {  // wave-uniform
            const int vk = l_var[k];
            if (vk < 0) continue;
            const unsigned long long ak = l_janc[k];
            if (is_j) {
              const int j = lane;
              double val = 0.0;
              if (((ak >> j) & 1ull) && my_rev && my_var >= 0) {
                val = my_jmul * l_jmul[k] * (my_a[0] * jcol[k * 3] + my_a[1] * jcol[k * 3 + 1] + my_a[2] * jcol[k * 3 + 2]);
                if (j != k && my_var == vk) val *= 2.0;
              }
              tmp[j] = val;
            }
            gen_sync();
            if (is_v) {
              double sum = 0.0;
              for (int e = l_famoff[lane]; e < l_famoff[lane + 1]; ++e) sum += tmp[l_fam[e]];
              if (sum != 0.0) {
                const int hi_ = lane > vk ? lane : vk, lo_ = lane > vk ? vk : lane;
                H[GEN_TRI(hi_, lo_)] += sum;
              }
            }
            gen_sync();
          }
        }
        GPROF_STAGE(3)  // second-order sweep
      };

      if (MODE == MODE_FK) {
        eval_value(x);
        if (is_t) {
          const int f = my_ft, row = tb.term_ref[lane];
          for (int i = 0; i < 3; ++i) kp.f64out[(it * kp.n_ref + row) * 3 + i] = P[f * 3 + i];
        }
        gen_sync();
        continue;
      }
      if (MODE == MODE_EVAL) {  // objective(x, grad): value without, gradient with the regulariser (quirk Q1)
        const double f = eval_value(x);
        assemble_model();
        if (lane == 0) {
          kp.f64out[r0] = f;
          if (dexpilot && kp.state) kp.state[r0] = nst;
        }
        if (is_v) kp.g64out[r0 * kp.n_opt + my_api] = g[lane] + 2.0 * delta * (x[lane] - my_xl);
        gen_sync();
        continue;
      }

      // ---- MODE_SOLVE ---------------------------------------------------------------------------------------------
      auto reg_at = [&](const double* xs) -> double {
        double p = 0.0;
        if (is_v) p = (xs[lane] - my_xl) * (xs[lane] - my_xl);
        return delta * gen_wave_sum(p);
      };
      auto add_reg_model = [&](const double* xs) {
        if (is_v) {
          g[lane] += 2.0 * delta * (xs[lane] - my_xl);
          H[GEN_TRI(lane, lane)] += 2.0 * delta;
        }
        gen_sync();
      };
      if (is_v) {
        x[lane] = fmin(fmax(x[lane], my_lo), my_hi);
        xt[lane] = x[lane];
      }
      gen_sync();
      // ONE value evaluation per pass (through ONE call site: every F that is ever compared comes from the same code, so the
      // comparison is not one of two copies' rounding) and the model -- gradient + Hessian, most of a pass -- only at points
      // that have been ACCEPTED: the kinematic state is the trial point's, so an accepted step goes straight on to
      // assemble_model(), a rejected one costs the kinematics and the terms alone.  The start point is "trial point 0".
      double F = 0.0, lam = (double)kp.lam0, nu = 2.0;
      int iters = 0, status = ST_MAXITER;
      const double tol = (double)kp.tol, cap = (double)kp.step_cap;
      bool bad = false, first = true;
      while (!bad && (first || iters < kp.max_iter)) {
        bool chol_ok = true;
        double smax = 0.0, pred = 0.0;
        GPROF_START();
        if (!first) {
          ++iters;
          GPROF_COUNT(11);
          // active set and damped system (lower triangle), lane = row
          if (is_v) {
            const bool a = (x[lane] <= my_lo && g[lane] > 0) || (x[lane] >= my_hi && g[lane] < 0);
            act[lane] = a ? 1.0 : 0.0;
          }
          gen_sync();
          GPROF_STAGE(4)  // active set
          chol_ok = gen_factor_solve<NV>(lane, nv, gen_lds_addr(H), gen_lds_addr(g), gen_lds_addr(act), lam, gen_lds_addr(Lt), gen_lds_addr(s));
          gen_sync();
          GPROF_STAGE(5)  // factorisation + triangular solves (registers)
          if (chol_ok) {
            GPROF_STAGE(6)
            double sm = is_v ? fabs(s[lane]) : 0.0;
            sm = gen_wave_max(sm);
            const double scale = (cap > 0 && sm > cap) ? RealTraits<double, true>::div(cap, sm) : 1.0;
            double my_s = 0.0, my_d = 0.0;
            bool clipped = false;
            if (is_v) {
              my_d = s[lane];
              const double xu = x[lane] + scale * my_d;
              const double xn = fmin(fmax(xu, my_lo), my_hi);
              clipped = xn != xu;
              xt[lane] = xn;
              my_s = xn - x[lane];
              s[lane] = my_s;
            }
            gen_sync();
            // predicted decrease -(g . s + s^T H s / 2) of the TAKEN step s
            double pp = 0.0;
            if (!__any(clipped)) {  // wave-uniform
              // no variable ran into its box: s = a d with (H_ff + lam I) d = -g_f, so the decrease follows from two sums --
              // a (1 - a / 2) (-g . d) + a^2 lam |d|^2 / 2 (variables held at a bound have d = 0) -- without touching H
              const double gd = gen_wave_sum(is_v ? -g[lane] * my_d : 0.0), dd = gen_wave_sum(my_d * my_d);
              pred = scale * (1.0 - 0.5 * scale) * gd + 0.5 * scale * scale * lam * dd;
            } else {
              // clipped: lane r sums its own row's strictly-lower part, s^T H s = sum_r s_r (2 sum_{u<r} H_ru s_u + H_rr s_r)
              if (is_v) {
                const double* Hr = H + GEN_TRI(lane, 0);
                double hs = 0.0;
                for (int u = 0; u < lane; ++u) hs += Hr[u] * s[u];
                pp = -(g[lane] * my_s + 0.5 * my_s * (2.0 * hs + Hr[lane] * my_s));
              }
              pred = gen_wave_sum(pp);
            }
            smax = gen_wave_max(fabs(my_s));
            GPROF_STAGE(7)  // step, predicted decrease
            if (lam <= (double)kp.lam0 && smax < (double)kp.blind_tol) {
              // an essentially undamped Newton step of a verified model shorter than blind_tol: its error is ~C s^2, far
              // below tol -- taken without a further evaluation (the rule of the specialised kernels)
              if (is_v) x[lane] = xt[lane];
              gen_sync();
              status = ST_CONVERGED;
              break;
            }
          }
        }
        bool accept = false;
        double Ft = 0.0;
        if (chol_ok) {
          Ft = eval_value(xt) + reg_at(xt);
          accept = first || ((Ft <= F) && (pred > 0));
        }
        if (first) {
          first = false;
          F = Ft;
          bad = !(F == F);
          if (!bad) {
            assemble_model();
            add_reg_model(xt);
          }
          continue;
        }
        if (accept) {
          const double rho = RealTraits<double, true>::div(F - Ft, fmax(pred, 1e-300));
          const bool small = smax < tol || pred <= 1e-18 * fmax(F, 1e-30);
          if (is_v) x[lane] = xt[lane];
          gen_sync();
          F = Ft;
          const double t3 = 2.0 * rho - 1.0;
          double shrink = fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3);
          if (kp.lam_fastdec > 0 && rho > 0.9) shrink = (double)kp.lam_fastdec;  // an accurate model: take the damping back fast
          lam = fmax(lam * shrink, 1e-12);
          nu = 2.0;
          if (small) {
            status = ST_CONVERGED;
            break;
          }
          assemble_model();  // the model of the point just accepted (its kinematics are still in LDS / registers)
          add_reg_model(x);
        } else {
          lam *= nu;
          nu *= 2.0;
          if (kp.lam_jump > 0.f) {
            // (round 6) go straight to a damping that matters next to the curvature -- the rule of every other family (lam_jump x
            // mean diagonal of the free block): creeping up from lambda0 = 1e-4 by x 2, x 4, x 8 ... an indefinite Newton model
            // took ten failed factorisations (each 40 % of a pass) to reach 1e-1; Shadow DexPilot on these tables needed 9.8
            // passes per frame where the sixteen-lane kernel needs 4.8
            double hd = 0.0, nf_ = 0.0;
            if (is_v && act[lane] == 0.0) {
              hd = H[GEN_TRI(lane, lane)];
              nf_ = 1.0;
            }
            hd = gen_wave_sum(hd);
            nf_ = gen_wave_sum(nf_);
            lam = fmax(lam, (double)kp.lam_jump * hd / fmax(nf_, 1.0));
          }
          if (lam > 1e12) {
            status = ST_CONVERGED;  // no descent direction left at any damping: x is stationary to rounding
            break;
          }
        }
        bad = !(F == F);
      }
      // ---- write the frame's answer ------------------------------------------------------------------------------------
      bool nonfinite = bad;
      if (is_v) nonfinite = nonfinite || !(x[lane] == x[lane]) || fabs(x[lane]) > 1e30;
      nonfinite = __any(nonfinite);
      if (nonfinite) status = ST_FALLBACK;  // like optimizer.py:100-102: last_qpos is returned
      if (is_v) {
        const double v = nonfinite ? my_xl : x[lane];
        if (nonfinite) x[lane] = v;
        kp.qout[it * ld + my_api] = (float)v;
        if (kp.qout64) kp.qout64[it * ld + my_api] = v;
      }
      if (lane == 0) {
        if (kp.status) kp.status[it] = status;
        if (kp.iters) kp.iters[it] = iters;
        if (kp.fval) kp.fval[it] = (float)F;
      }
      st_carry = nst;
      gen_sync();
    }
    if (MODE == MODE_SOLVE && dexpilot && kp.state && lane == 0) kp.state[r0] = st_carry;
    gen_sync();
  }
  GPROF_FLUSH()
}

#undef my_jmul
#undef my_lo
#undef my_hi
}  // namespace dexr
