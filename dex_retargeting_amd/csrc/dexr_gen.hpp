// dexr_gen.hpp -- the GENERAL solve kernel: any model the generic table format can describe (include/dexr_tables.h:
// up to 64 joints / variables / target links / reference rows in one component, any tree shape, mimic joints folded,
// 2..8 DexPilot fingers).  It exists so that nothing the reference's optimizers accept
// (/root/reference/src/dex_retargeting/optimizer.py:18-52: any URDF, any number of links and vectors) ends in a
// ValueError here; the specialised families (dexr_kernel / dexr_wide / dexr_red) stay the fast path for everything that
// fits them.
//
// Mapping: ONE WAVEFRONT PER FRAME (a block is one wave: barriers are free), float64 throughout, every table read from
// memory with rolled loops, all per-frame state in LDS:
//   joint values           lane k = joint k
//   forward kinematics     level-synchronous over the tree: the joints of depth d (one lane each) compose their parent's
//                          world transform (LDS) with their placement and motion -- depth, not joint count, steps
//   frames / terms         lane f = target link f; lane t = residual term t (SmoothL1 value, gradient, curvature)
//   gradient / Hessian     per term: lane k forms joint k's column a x (p - o), lane v folds its variable's joint family
//                          (kinematics_adaptor.py:102-113); the lower triangle of H is dealt out ENTRY by entry over the 64
//                          lanes (entry e = r (r + 1) / 2 + c on lane e mod 64) and accumulated over the terms in registers
//                          (round 4; round 3 gave lane v row v and read-modify-wrote it in LDS: the longest row set the
//                          pace); the second-order kinematic term is one sweep over the joints per pass from per-joint
//                          sums CF_k (round 3: a chain walk per term)
//   Cholesky / solves      lane = row, columns in sequence
// Solver: the projected Levenberg-Marquardt / Newton iteration on F = f + norm_delta |x - last|^2 that
// oracle/solvers.solve_lm_batched states (exact SmoothL1 curvature, second-order kinematic term, Nielsen damping), plus
// the trust radius the other kernels use.  What is computed per evaluation follows the reference's closures
// (optimizer.py:146-198, 249-304, 510-575) and the DexPilot pre-amble (:462-508).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dexr_kernel.hpp"

namespace dexr {

struct GenTab {  // device pointers into the uploaded generic table
  int32_t nj, nf, nt, nv, nfam, max_depth, has_kp;
  int32_t lt_in_lds;  // the backward substitution transposes the factor through a second packed triangle in LDS (see
                      // gen_factor_solve; chosen at model creation: create_generic in dexr_api.hip)
  const double *X, *axis, *jmul, *joff, *lo, *hi, *frame_off;
  const unsigned long long *frame_anc, *joint_anc;
  const int32_t *jtype, *parent, *depth, *src_idx, *var, *var_api, *fam_off, *fam, *frame_joint, *term_task, *term_origin,
      *term_ref, *row_ho, *row_ht;
};

// entries of the packed triangles in LDS: the Hessian and -- when the backward substitution transposes the factor through
// LDS (gen_factor_solve) -- the transposed factor
__host__ __device__ inline size_t gen_tri_doubles(int nv, bool lt_in_lds) {
  const size_t ntri = (size_t)nv * (nv + 1) / 2;
  return lt_in_lds ? 2 * ntri : ntri;
}

// doubles of LDS one wave needs
__host__ __device__ inline size_t gen_lds_doubles(int nj, int nf, int nt, int nv, int nfam, bool lt_in_lds) {
  const size_t state = (size_t)nj * 12 + nj * 3 + nj + (size_t)nf * 3 + (size_t)nt * 3 + nt + (size_t)nt * 3 + nt + (size_t)nt * 3 +
                       (size_t)nt * 3 + (size_t)nv * 6 + gen_tri_doubles(nv, lt_in_lds) + (size_t)nj * 3 + (size_t)nv * 3 + nj + 8;
  // wave-local copies of the tables the inner loops index (see "tables" in the kernel)
  const size_t ints = 3 * (size_t)nj + (size_t)nv + 1 + (size_t)nfam + (size_t)nf + 2 * (size_t)nt +
                      ((size_t)nv * (nv + 1) / 2 + 1) / 2 /* (row, column) of every entry of the packed triangle, 2 x uint8 */;
  return state + (size_t)nj /* jmul */ + (size_t)nf + (size_t)nj /* masks */ + (ints + 1) / 2;
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
// entries of the packed triangle per lane (accumulated in registers): 12 serves up to 38 variables (741 entries), 33 up to 64
constexpr int GEN_SLOTS_SMALL = 12, GEN_SLOTS_BIG = 33;

__device__ __forceinline__ double gen_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double gen_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}


// a double of lane `src` (a compile-time constant at every call site below) as a wave-uniform value: two v_readlane_b32
__device__ __forceinline__ double gen_readlane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Solve (H restricted to the free set + lam I) d = -g for the step d, IN REGISTERS: lane r holds row r of the lower
// triangle (NV doubles, statically indexed in fully unrolled loops), the factor is formed column by column
// (Cholesky-Crout): entry (r, c) = (A[r][c] - sum_{k<c} L[r][k] L[c][k]) / L[c][c] with row c's entries read from lane c
// (v_readlane with a constant lane: scalar operands, no LDS, no barrier).  Forward substitution column-wise (y_k broadcast
// from lane k); the backward substitution needs columns: see its two variants below.  Round 3 kept the matrix in LDS, lane = row, one block barrier + a
// read-modify-write sweep per column: 113 k of the 303 k cycles of a 37-variable pass, 26 k more for the solves.
// Not inlined: inside the kernel the register allocator carried ~400 live registers through the unrolled columns (92 on
// its own); as a call it costs 22 spilled registers at the call site.
// Returns false when a pivot is not positive (the damped model is indefinite: the caller raises lambda).
template <int NV>
__device__ __noinline__ bool gen_factor_solve(int lane, int nv, const double* H, const double* g, const double* act, double lam,
                                              double* Lt, double* s_out) {
  const bool rowv = lane < nv;
  const bool a_r = rowv ? act[lane] != 0.0 : true;
  const int base = lane * (lane + 1) / 2;
  double L[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    double v = 0.0;
    if (c < nv && rowv && c <= lane) {
      const bool au = a_r || act[c] != 0.0;
      v = au ? (c == lane ? 1.0 : 0.0) : H[base + c] + (c == lane ? lam : 0.0);
    }
    L[c] = v;
  }
  double rhs = (rowv && !a_r) ? -g[lane] : 0.0;
  double invd = 0.0;  // 1 / L[lane][lane]
  bool ok = true;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    if (c < nv && ok) {  // wave-uniform
      double acc = L[c];
#pragma unroll
      for (int k = 0; k < c; ++k) acc = fma(-L[k], gen_readlane(L[k], c), acc);  // L[k] of lane c = L[c][k]
      const double d = gen_readlane(acc, c);
      if (!(d > 0.0)) {
        ok = false;
      } else {
        // 1 / sqrt(d): hardware estimate + two Newton steps (full double precision for the well-scaled pivots of a damped
        // Hessian; a correctly rounded sqrt and a division cost three times the instructions, 38 times per solve)
        double ip = __builtin_amdgcn_rsq(d);
        ip = ip * fma(-0.5 * d * ip, ip, 1.5);
        ip = ip * fma(-0.5 * d * ip, ip, 1.5);
        L[c] = lane == c ? d * ip : (lane > c ? acc * ip : 0.0);
        if (lane == c) invd = ip;
      }
    }
  }
  if (!ok) return false;
  // L y = rhs: y_k is final on lane k once the columns before k have been subtracted
  double y = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (k < nv) {
      const double yk = gen_readlane(rhs * invd, k);
      if (lane == k) y = yk;
      if (lane > k) rhs = fma(-L[k], yk, rhs);
    }
  }
  // L^T s = y needs the COLUMNS of the factor, which no lane holds
  double sv = 0.0;
  bool in_regs = false;
  if constexpr (NV <= 38) in_regs = Lt == nullptr;  // wave-uniform
  if (in_regs) {
    // (38-variable instantiation, chosen per model when it buys a resident wave: an arm + hand's wave needs 20 KB of LDS
    // instead of 26 KB without the transposed copy -- 8 instead of 6 waves per CU, 29 -> 26 ms per 65 536 frames -- while
    // small models, whose LDS was not the limit, lose 10 % to the extra instructions.)  Every lane carries the whole
    // vector y as wave-uniform values (yv[k] = y_k read from lane k) and runs the substitution redundantly:
    // s_k = yv[k] / L[k][k], then yv[j] -= L[k][j] s_k with row k's entries read from lane k -- 2 v_readlane + 1 FMA per
    // entry like the factorisation, no LDS
    constexpr int NY = NV <= 38 ? NV : 1;
    double yv[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) yv[k] = k < nv ? gen_readlane(y, k) : 0.0;
#pragma unroll
    for (int k = NY - 1; k >= 0; --k) {
      if (k < nv) {
        const double sk = yv[k] * gen_readlane(invd, k);
        if (lane == k) sv = sk;
#pragma unroll
        for (int j = 0; j < k; ++j) yv[j] = fma(-gen_readlane(L[j], k), sk, yv[j]);
      }
    }
  } else {
    // the factor is transposed once through LDS (`Lt`, a second packed triangle), lane c then holds column c (entries
    // L[k][c], k >= c) and the substitution is column-wise like the forward one
    if (rowv) {
#pragma unroll
      for (int c = 0; c < NV; ++c)
        if (c < nv && c <= lane) Lt[base + c] = L[c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) L[k] = (k < nv && rowv && k >= lane) ? Lt[k * (k + 1) / 2 + lane] : 0.0;  // column `lane`
#pragma unroll
    for (int k = NV - 1; k >= 0; --k) {
      if (k < nv) {
        const double sk = gen_readlane(y * invd, k);
        if (lane == k) sv = sk;
        if (lane < k) y = fma(-L[k], sk, y);
      }
    }
  }
  if (rowv) s_out[lane] = sv;
  return true;
}

template <int MODE, int NSLOT>
__global__ void __launch_bounds__(64, NSLOT == GEN_SLOTS_SMALL ? 2 : 1) dexr_gen_kernel(KernelParams kp, GenTab tb) {
  extern __shared__ double gen_lds[];
  const int lane = threadIdx.x;
  const int nj = tb.nj, nf = tb.nf, nt = tb.nt, nv = tb.nv;
  double* Tw = gen_lds;            // nj x 12: world transform of every joint frame AFTER its motion (R row-major | p)
  double* aw = Tw + nj * 12;       // nj x 3: world axis
  double* qj = aw + nj * 3;        // nj
  double* P = qj + nj;             // nf x 3
  double* tgt = P + nf * 3;        // nt x 3
  double* wt = tgt + nt * 3;       // nt
  double* tg = wt + nt;            // nt x 3: dF/d(vector or position) of term t
  double* tc1 = tg + nt * 3;       // nt
  double* tc2 = tc1 + nt;          // nt x 3
  double* tu = tc2 + nt * 3;       // nt x 3
  double* x = tu + nt * 3;         // nv each:
  double* xl = x + nv;
  double* xt = xl + nv;
  double* g = xt + nv;
  double* s = g + nv;
  double* act = s + nv;
  const int ntri = nv * (nv + 1) / 2;
  double* H = act + nv;            // lower triangle, packed by rows: entry (r, c), c <= r, at r (r + 1) / 2 + c
  double* Lt = H + ntri;           // (tb.lt_in_lds only: the transposed factor, see gen_factor_solve)
  double* jcol = H + gen_tri_doubles(nv, tb.lt_in_lds != 0);  // nj x 3
  double* vcol = jcol + nj * 3;         // nv x 3
  double* tmp = vcol + nv * 3;          // nj
  double* flag = tmp + nj;              // 8 scalars
  // tables the inner loops index with data-dependent subscripts (families of a variable, parents along a chain,
  // ancestor masks): read from global memory they cost two dependent ~1 us round trips per access -- the first version of
  // this kernel spent ~90 % of a pass waiting for them -- so every wave keeps its own copy in LDS
  double* l_jmul = flag + 8;                                                      // nj
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
  uint16_t* l_rc = reinterpret_cast<uint16_t*>(l_torigin + nt);                   // ntri: row | column << 8 of entry e
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
  for (int e = lane; e < nv * (nv + 1) / 2; e += 64) {  // entry e of the packed triangle: row r, column c
    int r = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= e) ++r;  // (guards the rounding of the square root)
    while (r * (r + 1) / 2 > e) --r;
    l_rc[e] = (uint16_t)(r | ((e - r * (r + 1) / 2) << 8));
  }
  __syncthreads();

  // Lane k IS joint k (and variable k, frame k) for the whole kernel: what the tables say about them is read ONCE, into
  // registers.  (Round 3 re-read depth / placement / axis / offsets from global memory in every level of every forward
  // kinematics -- two dependent ~1 us round trips per level, 19 levels for an arm + hand, twice per pass: most of a pass.)
  const bool is_j = lane < nj, is_v = lane < nv, is_f = lane < nf;
  const int my_depth = is_j ? tb.depth[lane] : -1;
  const int my_src = is_j ? tb.src_idx[lane] : 0;
  double myX[12], my_ax[3], my_fo[3];
#pragma unroll
  for (int i = 0; i < 12; ++i) myX[i] = is_j ? tb.X[(size_t)lane * 12 + i] : 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    my_ax[i] = is_j ? tb.axis[lane * 3 + i] : 0.0;
    my_fo[i] = is_f ? tb.frame_off[(size_t)lane * 3 + i] : 0.0;
  }
  const double my_joff = is_j ? tb.joff[lane] : 0.0;
  const double my_lo = is_v ? tb.lo[lane] : 0.0, my_hi = is_v ? tb.hi[lane] : 0.0;
  const int my_api = is_v ? tb.var_api[lane] : 0;

  const bool no_mimic = tb.nfam == nv;  // every variable drives exactly one joint (families partition the driven joints)
  GPROF_DECL
  const int64_t cnt = kp.bucket ? (int64_t)kp.bucket[1] : kp.B;
  const int64_t boff = kp.bucket ? (int64_t)kp.bucket[0] : 0;
  const int ld = kp.ld;
  const bool seq = kp.T > 0;
  const int n_frames = seq ? kp.T : 1;
  const bool dexpilot = kp.kind == DEXR_KIND_DEXPILOT;
  const double beta = (double)kp.huber_delta, delta = (double)kp.norm_delta, inv_norm = (double)kp.inv_norm;

  for (int64_t item = blockIdx.x; item < cnt; item += gridDim.x) {
    const int64_t r0 = kp.perm ? (int64_t)kp.perm[boff + item] : item;
    uint32_t st_carry = 0u;
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
        xl[lane] = l;
      }
      uint32_t nst = 0;
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
          if (lane < nt) {
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
            for (int i = 0; i < 3; ++i) tgt[lane * 3 + i] = (double)tv[i];
            wt[lane] = (double)w;
          }
        } else if (lane < nt) {
          const float sc = (kp.kind == DEXR_KIND_VECTOR) ? kp.scaling : 1.f;
          float rv[3];
          ref_row(tb.term_ref[lane], rv);
          for (int i = 0; i < 3; ++i) tgt[lane * 3 + i] = (double)(rv[i] * sc);  // f32 multiply: optimizer.py:246
          wt[lane] = 1.0;
        }
      }
      __syncthreads();

      // ---- one evaluation at xs: kinematics, terms, value (eval_value); gradient + Hessian of the data term at the point
      // whose kinematic state is in LDS (assemble_model) ------------------------------------------------------------------
      auto eval_value = [&](const double* xs) -> double {
        GPROF_START();
        if (lane < nj) {
          const int v = l_var[lane];
          double q;
          if (MODE == MODE_FK) q = kp.xin[it * kp.n_q + my_src];
          else if (v >= 0) q = l_jmul[lane] * xs[v] + my_joff;
          else q = l_jmul[lane] * (double)kp.fixed[it * kp.ldf + my_src] + my_joff;
          qj[lane] = q;
        }
        __syncthreads();
        for (int d = 0; d <= tb.max_depth; ++d) {
          if (my_depth == d) {
            const int k = lane, pa = l_parent[k];
            double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
            if (pa >= 0) {
              for (int i = 0; i < 9; ++i) Rp[i] = Tw[pa * 12 + i];
              for (int i = 0; i < 3; ++i) pp[i] = Tw[pa * 12 + 9 + i];
            }
            const double* Xk = myX;
            double Ra[9], pa3[3];
            for (int i = 0; i < 3; ++i) {
              for (int j = 0; j < 3; ++j) Ra[3 * i + j] = Rp[3 * i] * Xk[j] + Rp[3 * i + 1] * Xk[3 + j] + Rp[3 * i + 2] * Xk[6 + j];
              pa3[i] = Rp[3 * i] * Xk[9] + Rp[3 * i + 1] * Xk[10] + Rp[3 * i + 2] * Xk[11] + pp[i];
            }
            const double ax = my_ax[0], ay = my_ax[1], az = my_ax[2];
            const double a0 = Ra[0] * ax + Ra[1] * ay + Ra[2] * az, a1 = Ra[3] * ax + Ra[4] * ay + Ra[5] * az,
                         a2 = Ra[6] * ax + Ra[7] * ay + Ra[8] * az;
            aw[k * 3] = a0; aw[k * 3 + 1] = a1; aw[k * 3 + 2] = a2;
            const double q = qj[k];
            if (l_jtype[k] == DEXR_JOINT_REVOLUTE) {  // Rodrigues about the local axis: I + sin K + (1 - cos) K^2
              double sn, cs;
              sincos_f64(q, &sn, &cs);  // (dexr_math.hpp: no slow path, unlike ocml's)
              const double c1 = 1.0 - cs;
              const double M[9] = {1 - c1 * (ay * ay + az * az), -sn * az + c1 * ax * ay, sn * ay + c1 * ax * az,
                                   sn * az + c1 * ax * ay, 1 - c1 * (ax * ax + az * az), -sn * ax + c1 * ay * az,
                                   -sn * ay + c1 * ax * az, sn * ax + c1 * ay * az, 1 - c1 * (ax * ax + ay * ay)};
              for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                  Tw[k * 12 + 3 * i + j] = Ra[3 * i] * M[j] + Ra[3 * i + 1] * M[3 + j] + Ra[3 * i + 2] * M[6 + j];
              for (int i = 0; i < 3; ++i) Tw[k * 12 + 9 + i] = pa3[i];
            } else {
              for (int i = 0; i < 9; ++i) Tw[k * 12 + i] = Ra[i];
              Tw[k * 12 + 9] = pa3[0] + a0 * q;
              Tw[k * 12 + 10] = pa3[1] + a1 * q;
              Tw[k * 12 + 11] = pa3[2] + a2 * q;
            }
          }
          __syncthreads();
        }
        if (lane < nf) {
          const int j = l_fjoint[lane];
          const double* o = my_fo;
          if (j < 0) {
            for (int i = 0; i < 3; ++i) P[lane * 3 + i] = o[i];
          } else {
            const double* T = Tw + j * 12;
            for (int i = 0; i < 3; ++i) P[lane * 3 + i] = T[3 * i] * o[0] + T[3 * i + 1] * o[1] + T[3 * i + 2] * o[2] + T[9 + i];
          }
        }
        __syncthreads();
        GPROF_STAGE(0)  // forward kinematics + frames
        if (MODE == MODE_FK) return 0.0;
        double fpart = 0.0;
        if (lane < nt) {
          const int ft = l_ttask[lane], fo = l_torigin[lane];
          double r[3];
          for (int i = 0; i < 3; ++i) r[i] = P[ft * 3 + i] - (fo >= 0 ? P[fo * 3 + i] : 0.0) - tgt[lane * 3 + i];
          if (kp.kind == DEXR_KIND_POSITION) {  // SmoothL1 per coordinate, mean over 3 P entries (optimizer.py:163-166)
            for (int i = 0; i < 3; ++i) {
              const double ad = fabs(r[i]);
              const bool in = ad < beta;
              fpart += (in ? 0.5 * r[i] * r[i] / beta : ad - 0.5 * beta) * inv_norm;
              tg[lane * 3 + i] = (in ? r[i] / beta : (r[i] > 0 ? 1.0 : (r[i] < 0 ? -1.0 : 0.0))) * inv_norm;
              tc2[lane * 3 + i] = in ? inv_norm / beta : 0.0;
            }
            tc1[lane] = 0.0;
          } else {  // SmoothL1 of the vector norm, weighted, mean over the V vectors (optimizer.py:262-273, 523-546)
            const double d = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            const bool in = d < beta;
            const double w = wt[lane];
            fpart = w * (in ? 0.5 * d * d / beta : d - 0.5 * beta) * inv_norm;
            const double psi = in ? d / beta : 1.0, kk = in ? 1.0 / beta : 0.0;
            const double gc = d > 0 ? w * psi * inv_norm / d : 0.0;  // torch.norm backward: zero at 0
            for (int i = 0; i < 3; ++i) {
              tg[lane * 3 + i] = gc * r[i];
              tu[lane * 3 + i] = d > 0 ? r[i] / d : 0.0;
            }
            tc1[lane] = gc;                          // H_vec = c1 (I - u u^T) + c2 u u^T
            tc2[lane * 3] = w * kk * inv_norm - gc;  // (c2 - c1)
          }
        }
        const double fval = gen_wave_sum(fpart);
        GPROF_STAGE(1)  // terms
        return fval;
      };
      auto assemble_model = [&]() {
        GPROF_START();
        if (lane < nv) g[lane] = 0.0;
        __syncthreads();
        double hacc[NSLOT];  // this lane's entries of H, accumulated over the terms (slot i: entry 64 i + lane)
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) hacc[i] = 0.0;
        double cf[3] = {0.0, 0.0, 0.0};  // lane k: CF_k (see below)
        int rc_[NSLOT];  // (row, column) of this lane's entries: read once per evaluation, not once per term
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) rc_[i] = (64 * i + lane < ntri) ? (int)l_rc[64 * i + lane] : 0;
        const bool pos = kp.kind == DEXR_KIND_POSITION;
        // entries of H owned by this lane += the contributions of term t (columns in vb, u . column in ub) and -- `two` -- of
        // term t + 1 (vb2, ub2), formed in one body so that both terms' LDS reads are in flight together
        auto add_entries = [&](int t, const double* vb, const double* ub, bool two, const double* vb2, const double* ub2) {
          const int t2 = two ? t + 1 : t;
          const double k0 = pos ? tc2[t * 3] : tc1[t], k1 = pos ? tc2[t * 3 + 1] : tc1[t], k2 = pos ? tc2[t * 3 + 2] : tc1[t];
          const double kb = pos ? 0.0 : tc2[t * 3];
          const double m0 = pos ? tc2[t2 * 3] : tc1[t2], m1 = pos ? tc2[t2 * 3 + 1] : tc1[t2], m2 = pos ? tc2[t2 * 3 + 2] : tc1[t2];
          const double mb = pos ? 0.0 : tc2[t2 * 3];
#pragma unroll
          for (int i4 = 0; i4 < NSLOT; i4 += 4) {
            if (64 * i4 < ntri) {  // wave-uniform, per group of four slots: the slots of a group overlap their LDS reads
#pragma unroll
              for (int i = i4; i < i4 + 4 && i < NSLOT; ++i) {
                if (64 * i + lane < ntri) {
                  const int r = rc_[i] & 0xFF, c = rc_[i] >> 8;
                  // position: sum_i k_i v_r[i] v_c[i]  |  vector kinds: c1 (v_r . v_c) + (c2 - c1) (u . v_r) (u . v_c)
                  double add = k0 * vb[r * 3] * vb[c * 3] + k1 * vb[r * 3 + 1] * vb[c * 3 + 1] + k2 * vb[r * 3 + 2] * vb[c * 3 + 2];
                  if (!pos) add += kb * ub[r] * ub[c];  // (0 x uninitialised LDS is not 0)
                  if (two) {
                    add += m0 * vb2[r * 3] * vb2[c * 3] + m1 * vb2[r * 3 + 1] * vb2[c * 3 + 1] + m2 * vb2[r * 3 + 2] * vb2[c * 3 + 2];
                    if (!pos) add += mb * ub2[r] * ub2[c];
                  }
                  hacc[i] += add;
                }
              }
            }
          }
        };
        // joint k's column of term t: a x (p - o) over the term's two frames (sign), or the axis (prismatic)
        auto joint_column = [&](int t, double (&c)[3]) {
          const int ft = l_ttask[t], fo = l_torigin[t];
          const unsigned long long at = l_fanc[ft], ao = fo >= 0 ? l_fanc[fo] : 0ull;
          const int k = lane;
          c[0] = 0.0; c[1] = 0.0; c[2] = 0.0;
          const bool rev = l_jtype[k] == DEXR_JOINT_REVOLUTE;
          for (int side = 0; side < 2; ++side) {
            const bool on = ((side ? ao : at) >> k) & 1ull;
            if (!on) continue;
            const int fr = side ? fo : ft;
            const double sg = side ? -1.0 : 1.0;
            if (rev) {
              const double dx = P[fr * 3] - Tw[k * 12 + 9], dy = P[fr * 3 + 1] - Tw[k * 12 + 10], dz = P[fr * 3 + 2] - Tw[k * 12 + 11];
              c[0] += sg * (aw[k * 3 + 1] * dz - aw[k * 3 + 2] * dy);
              c[1] += sg * (aw[k * 3 + 2] * dx - aw[k * 3] * dz);
              c[2] += sg * (aw[k * 3] * dy - aw[k * 3 + 1] * dx);
            } else {
              for (int i = 0; i < 3; ++i) c[i] += sg * aw[k * 3 + i];
            }
          }
          // second-order kinematic term, first half: CF_k = sum over the terms of (column of joint k) x (force of the
          // term) -- the entry for a pair (joint k, revolute ancestor-or-self j) is m_j m_k a_j . CF_k, formed ONCE per
          // pass after the term loop (tg . (a_j x c) = a_j . (c x tg))
          cf[0] += c[1] * tg[t * 3 + 2] - c[2] * tg[t * 3 + 1];
          cf[1] += c[2] * tg[t * 3] - c[0] * tg[t * 3 + 2];
          cf[2] += c[0] * tg[t * 3 + 1] - c[1] * tg[t * 3];
        };
        if (no_mimic) {
          // every variable drives exactly one joint: lane k writes its variable's column itself, the columns are
          // double-buffered (vcol | jcol, tmp | act: both spare here) -- ONE barrier per term instead of three
          const int v = lane < nj ? l_var[lane] : -1;
          const double m = lane < nj ? l_jmul[lane] : 0.0;
          // lane k: its variable's column of term t into the buffer pair (vb, ub), its gradient entry
          auto own_column = [&](int t, double* vb, double* ub) {
            double c[3];
            joint_column(t, c);
            if (v >= 0) {
              const double c0 = m * c[0], c1 = m * c[1], c2 = m * c[2];
              vb[v * 3] = c0; vb[v * 3 + 1] = c1; vb[v * 3 + 2] = c2;
              g[v] += tg[t * 3] * c0 + tg[t * 3 + 1] * c1 + tg[t * 3 + 2] * c2;
              if (!pos) ub[v] = tu[t * 3] * c0 + tu[t * 3 + 1] * c1 + tu[t * 3 + 2] * c2;
            }
          };
          // terms in PAIRS: both terms' columns are formed and published before one barrier, both terms' entries are added
          // before the next -- as many barriers per term as the double-buffered single-term loop had (one), but two
          // independent chains of LDS round trips in flight in either phase
          for (int t = 0; t < nt; t += 2) {
            const bool two = t + 1 < nt;  // wave-uniform
            if (lane < nj) {
              own_column(t, vcol, tmp);
              if (two) own_column(t + 1, jcol, act);
            }
            __syncthreads();
            add_entries(t, vcol, tmp, two, jcol, act);
            __syncthreads();
          }
        } else {
          for (int t = 0; t < nt; ++t) {
            if (lane < nj) {
              double c[3];
              joint_column(t, c);
              for (int i = 0; i < 3; ++i) jcol[lane * 3 + i] = c[i];
            }
            __syncthreads();
            if (lane < nv) {  // the variable's column: its joint family folded (kinematics_adaptor.py:102-113)
              double c[3] = {0, 0, 0};
              for (int e = l_famoff[lane]; e < l_famoff[lane + 1]; ++e) {
                const int k = l_fam[e];
                const double mk = l_jmul[k];
                for (int i = 0; i < 3; ++i) c[i] += mk * jcol[k * 3 + i];
              }
              for (int i = 0; i < 3; ++i) vcol[lane * 3 + i] = c[i];
              g[lane] += tg[t * 3] * c[0] + tg[t * 3 + 1] * c[1] + tg[t * 3 + 2] * c[2];
              // u . column (vector kinds only: the position objective has no unit vector, its tu block is never written)
              if (!pos) tmp[lane] = tu[t * 3] * c[0] + tu[t * 3 + 1] * c[1] + tu[t * 3 + 2] * c[2];
            }
            __syncthreads();
            add_entries(t, vcol, tmp, false, vcol, tmp);
            __syncthreads();
          }
        }
        GPROF_STAGE(2)  // term loop: columns, gradient, Hessian entries
        // second-order kinematic term, second half: for every joint k that moves with a variable and every revolute
        // ancestor-or-self j of k that does too,  H[var j][var k] += m_j m_k a_j . CF_k  (twice for j != k inside one
        // family: both orders of the unordered pair).
        if (kp.newton) {
          if (lane < nj) {
            for (int i = 0; i < 3; ++i) jcol[lane * 3 + i] = cf[i];
          }
          __syncthreads();
        }
        if (kp.newton && no_mimic) {
          // one joint per variable: the owner of entry (r, c) forms its (at most two) contributions itself -- joint of r as
          // the moved joint with the joint of c as its ancestor, and the other way round
#pragma unroll
          for (int i = 0; i < NSLOT; ++i) {
            if (64 * i < ntri) {
              if (64 * i + lane < ntri) {
                const int r = rc_[i] & 0xFF, c = rc_[i] >> 8;
                const int jr = l_fam[l_famoff[r]], jc = l_fam[l_famoff[c]];
                double val = 0.0;
                if (((l_janc[jr] >> jc) & 1ull) && l_jtype[jc] == DEXR_JOINT_REVOLUTE)  // k = jr, j = jc
                  val += l_jmul[jc] * l_jmul[jr] * (aw[jc * 3] * jcol[jr * 3] + aw[jc * 3 + 1] * jcol[jr * 3 + 1] + aw[jc * 3 + 2] * jcol[jr * 3 + 2]);
                if (jr != jc && ((l_janc[jc] >> jr) & 1ull) && l_jtype[jr] == DEXR_JOINT_REVOLUTE)  // k = jc, j = jr
                  val += l_jmul[jr] * l_jmul[jc] * (aw[jr * 3] * jcol[jc * 3] + aw[jr * 3 + 1] * jcol[jc * 3 + 1] + aw[jr * 3 + 2] * jcol[jc * 3 + 2]);
                hacc[i] += val;
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < NSLOT; ++i)
          if (64 * i + lane < ntri) H[64 * i + lane] = hacc[i];
        __syncthreads();
        if (kp.newton && !no_mimic) {
          // families of several joints: one sweep over the joints per PASS -- lane j forms the entry of pair (j, k), lane v
          // sums its variable's family and adds into its own entries.  (Round 3 walked the chains of every TERM instead --
          // terms x chain depth x 2 block barriers, ~1 500 per pass for an arm + hand: most of the ~300 us a pass took.)
          for (int k = 0; k < nj; ++k) {  // wave-uniform
            const int vk = l_var[k];
            if (vk < 0) continue;
            const unsigned long long ak = l_janc[k];
            if (lane < nj) {
              const int j = lane;
              double val = 0.0;
              if (((ak >> j) & 1ull) && l_jtype[j] == DEXR_JOINT_REVOLUTE && l_var[j] >= 0) {
                val = l_jmul[j] * l_jmul[k] * (aw[j * 3] * jcol[k * 3] + aw[j * 3 + 1] * jcol[k * 3 + 1] + aw[j * 3 + 2] * jcol[k * 3 + 2]);
                if (j != k && l_var[j] == vk) val *= 2.0;
              }
              tmp[j] = val;
            }
            __syncthreads();
            if (lane < nv) {
              double sum = 0.0;
              for (int e = l_famoff[lane]; e < l_famoff[lane + 1]; ++e) sum += tmp[l_fam[e]];
              if (sum != 0.0) {
                const int hi_ = lane > vk ? lane : vk, lo_ = lane > vk ? vk : lane;
                H[GEN_TRI(hi_, lo_)] += sum;
              }
            }
            __syncthreads();
          }
        }
        GPROF_STAGE(3)  // second-order sweep
      };

      if (MODE == MODE_FK) {
        eval_value(x);
        if (lane < nt) {
          const int f = tb.term_task[lane], row = tb.term_ref[lane];
          for (int i = 0; i < 3; ++i) kp.f64out[(it * kp.n_ref + row) * 3 + i] = P[f * 3 + i];
        }
        __syncthreads();
        continue;
      }
      if (MODE == MODE_EVAL) {  // objective(x, grad): value without, gradient with the regulariser (quirk Q1)
        const double f = eval_value(x);
        assemble_model();
        if (lane == 0) {
          kp.f64out[r0] = f;
          if (dexpilot && kp.state) kp.state[r0] = nst;
        }
        if (lane < nv) kp.g64out[r0 * kp.n_opt + my_api] = g[lane] + 2.0 * delta * (x[lane] - xl[lane]);
        __syncthreads();
        continue;
      }

      // ---- MODE_SOLVE ---------------------------------------------------------------------------------------------
      auto reg_at = [&](const double* xs) -> double {
        double p = 0.0;
        if (lane < nv) p = (xs[lane] - xl[lane]) * (xs[lane] - xl[lane]);
        return delta * gen_wave_sum(p);
      };
      auto add_reg_model = [&](const double* xs) {
        if (lane < nv) {
          g[lane] += 2.0 * delta * (xs[lane] - xl[lane]);
          H[GEN_TRI(lane, lane)] += 2.0 * delta;
        }
        __syncthreads();
      };
      if (lane < nv) {
        x[lane] = fmin(fmax(x[lane], my_lo), my_hi);
        xt[lane] = x[lane];
      }
      __syncthreads();
      // ONE value evaluation per pass (through ONE call site: every F that is ever compared comes from the same code, so the
      // comparison is not one of two copies' rounding) and the model -- gradient + Hessian, most of a pass -- only at points
      // that have been ACCEPTED: the kinematic state in LDS is the trial point's, so an accepted step goes straight on to
      // assemble_model(), a rejected one costs the kinematics and the terms alone.  (Round 3: value at the trial point,
      // then value + model again at the same point through a second inlined copy.)  The start point is "trial point 0".
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
          if (lane < nv) {
            const bool a = (x[lane] <= my_lo && g[lane] > 0) || (x[lane] >= my_hi && g[lane] < 0);
            act[lane] = a ? 1.0 : 0.0;
          }
          __syncthreads();
          GPROF_STAGE(4)  // active set
          constexpr int NV = NSLOT == GEN_SLOTS_SMALL ? 38 : 64;  // rows the register factorisation holds
          chol_ok = gen_factor_solve<NV>(lane, nv, H, g, act, lam, tb.lt_in_lds ? Lt : nullptr, s);
          __syncthreads();
          GPROF_STAGE(5)  // factorisation + triangular solves (registers)
          if (chol_ok) {
            GPROF_STAGE(6)
            double sm = lane < nv ? fabs(s[lane]) : 0.0;
            sm = gen_wave_max(sm);
            const double scale = (cap > 0 && sm > cap) ? cap / sm : 1.0;
            if (lane < nv) {
              const double xn = fmin(fmax(x[lane] + scale * s[lane], my_lo), my_hi);
              xt[lane] = xn;
              s[lane] = xn - x[lane];
            }
            __syncthreads();
            double pp = 0.0, am = 0.0;
            if (lane < nv) {
              double hs = 0.0;
              for (int u = 0; u < nv; ++u) hs += (u <= lane ? H[GEN_TRI(lane, u)] : H[GEN_TRI(u, lane)]) * s[u];
              pp = -(g[lane] * s[lane] + 0.5 * s[lane] * hs);
              am = fabs(s[lane]);
            }
            pred = gen_wave_sum(pp);
            smax = gen_wave_max(am);
            GPROF_STAGE(7)  // step, predicted decrease
            if (lam <= (double)kp.lam0 && smax < (double)kp.blind_tol) {
              // an essentially undamped Newton step of a verified model shorter than blind_tol: its error is ~C s^2, far
              // below tol -- taken without a further evaluation (the rule of the specialised kernels)
              if (lane < nv) x[lane] = xt[lane];
              __syncthreads();
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
          const double rho = (F - Ft) / fmax(pred, 1e-300);
          const bool small = smax < tol || pred <= 1e-18 * fmax(F, 1e-30);
          if (lane < nv) x[lane] = xt[lane];
          __syncthreads();
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
          assemble_model();  // the model of the point just accepted (its kinematics are still in LDS)
          add_reg_model(x);
        } else {
          lam *= nu;
          nu *= 2.0;
          if (lam > 1e12) {
            status = ST_CONVERGED;  // no descent direction left at any damping: x is stationary to rounding
            break;
          }
        }
        bad = !(F == F);
      }
      // ---- write the frame's answer ------------------------------------------------------------------------------------
      bool nonfinite = bad;
      if (lane < nv) nonfinite = nonfinite || !(x[lane] == x[lane]) || fabs(x[lane]) > 1e30;
      nonfinite = __any(nonfinite);
      if (nonfinite) status = ST_FALLBACK;  // like optimizer.py:100-102: last_qpos is returned
      if (lane < nv) {
        const double v = nonfinite ? xl[lane] : x[lane];
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
      __syncthreads();
    }
    if (MODE == MODE_SOLVE && dexpilot && kp.state && lane == 0) kp.state[r0] = st_carry;
    __syncthreads();
  }
  GPROF_FLUSH()
}

}  // namespace dexr
