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
//                          (kinematics_adaptor.py:102-113) and accumulates ITS ROW of the lower triangle of H; the
//                          second-order kinematic term walks the term's chains, lane j = revolute ancestor j
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
  int32_t nj, nf, nt, nv, nfam, max_depth, has_kp, pad_;
  const double *X, *axis, *jmul, *joff, *lo, *hi, *frame_off;
  const unsigned long long *frame_anc, *joint_anc;
  const int32_t *jtype, *parent, *depth, *src_idx, *var, *var_api, *fam_off, *fam, *frame_joint, *term_task, *term_origin,
      *term_ref, *row_ho, *row_ht;
};

// doubles of LDS one wave needs
__host__ __device__ inline size_t gen_lds_doubles(int nj, int nf, int nt, int nv, int nfam) {
  const size_t state = (size_t)nj * 12 + nj * 3 + nj + (size_t)nf * 3 + (size_t)nt * 3 + nt + (size_t)nt * 3 + nt + (size_t)nt * 3 +
                       (size_t)nt * 3 + (size_t)nv * 8 + 2 * (size_t)nv * nv + (size_t)nj * 3 + (size_t)nv * 3 + nj + 8;
  // wave-local copies of the tables the inner loops index (see "tables" in the kernel)
  const size_t ints = 3 * (size_t)nj + (size_t)nv + 1 + (size_t)nfam + (size_t)nf + 2 * (size_t)nt;
  return state + (size_t)nj /* jmul */ + (size_t)nf + (size_t)nj /* masks */ + (ints + 1) / 2;
}

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

template <int MODE>
__global__ void __launch_bounds__(64) dexr_gen_kernel(KernelParams kp, GenTab tb) {
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
  double* rhs = s + nv;
  double* act = rhs + nv;
  double* ybuf = act + nv;
  double* H = ybuf + nv;           // nv x nv, lower triangle used
  double* Hf = H + (size_t)nv * nv;
  double* jcol = Hf + (size_t)nv * nv;  // nj x 3
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
        const int api = tb.var_api[lane];
        double v, l;
        if (MODE == MODE_EVAL) v = kp.xin[it * kp.n_opt + api];
        else if (carry) v = (double)(float)x[lane];  // the reference carries the float32 result (optimizer.py:99)
        else if (kp.x0) v = (double)kp.x0[r0 * ld + api];
        else v = (double)kp.last[r0 * ld + api];
        l = carry ? v : (double)kp.last[r0 * ld + api];
        if (seq) {  // seq_retarget.py:118-120: last_qpos clipped to the joint limits before every solve
          l = fmin(fmax(l, tb.lo[lane] + (double)kp.clip_eps), tb.hi[lane] - (double)kp.clip_eps);
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

      // ---- one evaluation at xs: kinematics, terms, value; with `model` also gradient + Hessian of F -------------------
      auto eval_at = [&](const double* xs, bool model) -> double {
        if (lane < nj) {
          const int v = l_var[lane];
          double q;
          if (MODE == MODE_FK) q = kp.xin[it * kp.n_q + tb.src_idx[lane]];
          else if (v >= 0) q = l_jmul[lane] * xs[v] + tb.joff[lane];
          else q = l_jmul[lane] * (double)kp.fixed[it * kp.ldf + tb.src_idx[lane]] + tb.joff[lane];
          qj[lane] = q;
        }
        __syncthreads();
        for (int d = 0; d <= tb.max_depth; ++d) {
          if (lane < nj && tb.depth[lane] == d) {
            const int k = lane, pa = l_parent[k];
            double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
            if (pa >= 0) {
              for (int i = 0; i < 9; ++i) Rp[i] = Tw[pa * 12 + i];
              for (int i = 0; i < 3; ++i) pp[i] = Tw[pa * 12 + 9 + i];
            }
            const double* Xk = tb.X + (size_t)k * 12;
            double Ra[9], pa3[3];
            for (int i = 0; i < 3; ++i) {
              for (int j = 0; j < 3; ++j) Ra[3 * i + j] = Rp[3 * i] * Xk[j] + Rp[3 * i + 1] * Xk[3 + j] + Rp[3 * i + 2] * Xk[6 + j];
              pa3[i] = Rp[3 * i] * Xk[9] + Rp[3 * i + 1] * Xk[10] + Rp[3 * i + 2] * Xk[11] + pp[i];
            }
            const double ax = tb.axis[k * 3], ay = tb.axis[k * 3 + 1], az = tb.axis[k * 3 + 2];
            const double a0 = Ra[0] * ax + Ra[1] * ay + Ra[2] * az, a1 = Ra[3] * ax + Ra[4] * ay + Ra[5] * az,
                         a2 = Ra[6] * ax + Ra[7] * ay + Ra[8] * az;
            aw[k * 3] = a0; aw[k * 3 + 1] = a1; aw[k * 3 + 2] = a2;
            const double q = qj[k];
            if (l_jtype[k] == DEXR_JOINT_REVOLUTE) {  // Rodrigues about the local axis: I + sin K + (1 - cos) K^2
              double sn, cs;
              sincos(q, &sn, &cs);
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
          const double* o = tb.frame_off + (size_t)lane * 3;
          if (j < 0) {
            for (int i = 0; i < 3; ++i) P[lane * 3 + i] = o[i];
          } else {
            const double* T = Tw + j * 12;
            for (int i = 0; i < 3; ++i) P[lane * 3 + i] = T[3 * i] * o[0] + T[3 * i + 1] * o[1] + T[3 * i + 2] * o[2] + T[9 + i];
          }
        }
        __syncthreads();
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
        if (!model) return fval;
        if (lane < nv) {
          g[lane] = 0.0;
          for (int u = 0; u <= lane; ++u) H[(size_t)lane * nv + u] = 0.0;
        }
        __syncthreads();
        double cf[3] = {0.0, 0.0, 0.0};  // lane k: CF_k (see below)
        for (int t = 0; t < nt; ++t) {
          const int ft = l_ttask[t], fo = l_torigin[t];
          const unsigned long long at = l_fanc[ft], ao = fo >= 0 ? l_fanc[fo] : 0ull;
          if (lane < nj) {
            const int k = lane;
            double c[3] = {0, 0, 0};
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
            for (int i = 0; i < 3; ++i) jcol[k * 3 + i] = c[i];
            // second-order kinematic term, first half: CF_k = sum over the terms of (column of joint k) x (force of the
            // term) -- the entry for a pair (joint k, revolute ancestor-or-self j) is m_j m_k a_j . CF_k, formed ONCE per
            // pass after this loop (tg . (a_j x c) = a_j . (c x tg))
            cf[0] += c[1] * tg[t * 3 + 2] - c[2] * tg[t * 3 + 1];
            cf[1] += c[2] * tg[t * 3] - c[0] * tg[t * 3 + 2];
            cf[2] += c[0] * tg[t * 3 + 1] - c[1] * tg[t * 3];
          }
          __syncthreads();
          if (lane < nv) {
            double c[3] = {0, 0, 0};
            for (int e = l_famoff[lane]; e < l_famoff[lane + 1]; ++e) {
              const int k = l_fam[e];
              const double m = l_jmul[k];
              for (int i = 0; i < 3; ++i) c[i] += m * jcol[k * 3 + i];
            }
            for (int i = 0; i < 3; ++i) vcol[lane * 3 + i] = c[i];
            g[lane] += tg[t * 3] * c[0] + tg[t * 3 + 1] * c[1] + tg[t * 3 + 2] * c[2];
          }
          __syncthreads();
          if (lane < nv) {
            const double c0 = vcol[lane * 3], c1v = vcol[lane * 3 + 1], c2v = vcol[lane * 3 + 2];
            if (kp.kind == DEXR_KIND_POSITION) {
              const double k0 = tc2[t * 3] * c0, k1 = tc2[t * 3 + 1] * c1v, k2 = tc2[t * 3 + 2] * c2v;
              for (int u = 0; u <= lane; ++u)
                H[(size_t)lane * nv + u] += k0 * vcol[u * 3] + k1 * vcol[u * 3 + 1] + k2 * vcol[u * 3 + 2];
            } else {
              const double a = tc1[t], b = tc2[t * 3];
              const double uv = tu[t * 3] * c0 + tu[t * 3 + 1] * c1v + tu[t * 3 + 2] * c2v;
              for (int u = 0; u <= lane; ++u) {
                const double uu = tu[t * 3] * vcol[u * 3] + tu[t * 3 + 1] * vcol[u * 3 + 1] + tu[t * 3 + 2] * vcol[u * 3 + 2];
                H[(size_t)lane * nv + u] += a * (c0 * vcol[u * 3] + c1v * vcol[u * 3 + 1] + c2v * vcol[u * 3 + 2]) + b * uv * uu;
              }
            }
          }
          __syncthreads();
        }
        if (kp.newton) {
          // second-order kinematic term, second half: for every joint k that moves with a variable and every revolute
          // ancestor-or-self j of k that does too,  H[var j][var k] += m_j m_k a_j . CF_k  (twice for j != k inside one
          // family: both orders of the unordered pair).  One sweep over the joints per PASS: lane j forms the entry of
          // pair (j, k), lane v sums its variable's family and adds into its own entries.  (Round 3 walked the chains of
          // every TERM instead -- terms x chain depth x 2 block barriers, ~1 500 per pass for an arm + hand: it was most of
          // the ~300 us a pass of that model took.)
          if (lane < nj) {
            for (int i = 0; i < 3; ++i) jcol[lane * 3 + i] = cf[i];
          }
          __syncthreads();
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
                H[(size_t)hi_ * nv + lo_] += sum;
              }
            }
            __syncthreads();
          }
        }
        return fval;
      };

      if (MODE == MODE_FK) {
        eval_at(x, false);
        if (lane < nt) {
          const int f = tb.term_task[lane], row = tb.term_ref[lane];
          for (int i = 0; i < 3; ++i) kp.f64out[(it * kp.n_ref + row) * 3 + i] = P[f * 3 + i];
        }
        __syncthreads();
        continue;
      }
      if (MODE == MODE_EVAL) {  // objective(x, grad): value without, gradient with the regulariser (quirk Q1)
        const double f = eval_at(x, true);
        if (lane == 0) {
          kp.f64out[r0] = f;
          if (dexpilot && kp.state) kp.state[r0] = nst;
        }
        if (lane < nv) kp.g64out[r0 * kp.n_opt + tb.var_api[lane]] = g[lane] + 2.0 * delta * (x[lane] - xl[lane]);
        __syncthreads();
        continue;
      }

      // ---- MODE_SOLVE ---------------------------------------------------------------------------------------------
      auto reg_at = [&](const double* xs) -> double {
        double p = 0.0;
        if (lane < nv) p = (xs[lane] - xl[lane]) * (xs[lane] - xl[lane]);
        return delta * gen_wave_sum(p);
      };
      auto add_reg_model = [&]() {
        if (lane < nv) {
          g[lane] += 2.0 * delta * (x[lane] - xl[lane]);
          H[(size_t)lane * nv + lane] += 2.0 * delta;
        }
        __syncthreads();
      };
      if (lane < nv) x[lane] = fmin(fmax(x[lane], tb.lo[lane]), tb.hi[lane]);
      __syncthreads();
      double F = eval_at(x, true) + reg_at(x);
      add_reg_model();
      double lam = (double)kp.lam0, nu = 2.0;
      int iters = 0, status = ST_MAXITER;
      const double tol = (double)kp.tol, cap = (double)kp.step_cap;
      bool bad = !(F == F);
      while (!bad && iters < kp.max_iter) {
        ++iters;
        // active set and damped system (lower triangle), lane = row
        if (lane < nv) {
          const bool a = (x[lane] <= tb.lo[lane] && g[lane] > 0) || (x[lane] >= tb.hi[lane] && g[lane] < 0);
          act[lane] = a ? 1.0 : 0.0;
        }
        __syncthreads();
        if (lane < nv) {
          const bool a = act[lane] != 0.0;
          for (int u = 0; u <= lane; ++u) {
            const bool au = a || act[u] != 0.0;
            Hf[(size_t)lane * nv + u] = au ? (u == lane ? 1.0 : 0.0) : H[(size_t)lane * nv + u] + (u == lane ? lam : 0.0);
          }
          rhs[lane] = a ? 0.0 : -g[lane];
        }
        __syncthreads();
        bool chol_ok = true;
        for (int p = 0; p < nv; ++p) {
          if (lane == p) {
            const double dd = Hf[(size_t)p * nv + p];
            flag[0] = dd > 0 ? sqrt(dd) : -1.0;
          }
          __syncthreads();
          const double piv = flag[0];
          if (!(piv > 0)) {
            chol_ok = false;
            break;
          }
          if (lane == p) Hf[(size_t)p * nv + p] = piv;
          if (lane > p && lane < nv) Hf[(size_t)lane * nv + p] /= piv;
          __syncthreads();
          if (lane > p && lane < nv) {
            const double lip = Hf[(size_t)lane * nv + p];
            for (int j = p + 1; j <= lane; ++j) Hf[(size_t)lane * nv + j] -= lip * Hf[(size_t)j * nv + p];
          }
          __syncthreads();
        }
        bool accept = false;
        double smax = 0.0, pred = 0.0, Ft = 0.0;
        if (chol_ok) {
          for (int p = 0; p < nv; ++p) {  // L y = rhs
            if (lane == p) ybuf[p] = rhs[p] / Hf[(size_t)p * nv + p];
            __syncthreads();
            if (lane > p && lane < nv) rhs[lane] -= Hf[(size_t)lane * nv + p] * ybuf[p];
            __syncthreads();
          }
          for (int p = nv - 1; p >= 0; --p) {  // L^T s = y
            if (lane == p) s[p] = ybuf[p] / Hf[(size_t)p * nv + p];
            __syncthreads();
            if (lane < p) ybuf[lane] -= Hf[(size_t)p * nv + lane] * s[p];
            __syncthreads();
          }
          double sm = lane < nv ? fabs(s[lane]) : 0.0;
          sm = gen_wave_max(sm);
          const double scale = (cap > 0 && sm > cap) ? cap / sm : 1.0;
          if (lane < nv) {
            const double xn = fmin(fmax(x[lane] + scale * s[lane], tb.lo[lane]), tb.hi[lane]);
            xt[lane] = xn;
            s[lane] = xn - x[lane];
          }
          __syncthreads();
          double pp = 0.0, am = 0.0;
          if (lane < nv) {
            double hs = 0.0;
            for (int u = 0; u < nv; ++u) hs += (u <= lane ? H[(size_t)lane * nv + u] : H[(size_t)u * nv + lane]) * s[u];
            pp = -(g[lane] * s[lane] + 0.5 * s[lane] * hs);
            am = fabs(s[lane]);
          }
          pred = gen_wave_sum(pp);
          smax = gen_wave_max(am);
          if (lam <= (double)kp.lam0 && smax < (double)kp.blind_tol) {
            // an essentially undamped Newton step of a verified model shorter than blind_tol: its error is ~C s^2, far
            // below tol -- taken without a further evaluation (the rule of the specialised kernels)
            if (lane < nv) x[lane] = xt[lane];
            __syncthreads();
            status = ST_CONVERGED;
            break;
          }
          Ft = eval_at(xt, false) + reg_at(xt);
          accept = (Ft <= F) && (pred > 0);
        }
        if (accept) {
          const double rho = (F - Ft) / fmax(pred, 1e-300);
          const bool small = smax < tol || pred <= 1e-18 * fmax(F, 1e-30);
          if (lane < nv) x[lane] = xt[lane];
          __syncthreads();
          F = eval_at(x, true) + reg_at(x);
          add_reg_model();
          const double t3 = 2.0 * rho - 1.0;
          double shrink = fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3);
          if (kp.lam_fastdec > 0 && rho > 0.9) shrink = (double)kp.lam_fastdec;  // an accurate model: take the damping back fast
          lam = fmax(lam * shrink, 1e-12);
          nu = 2.0;
          if (small) {
            status = ST_CONVERGED;
            break;
          }
        } else {
          lam *= nu;
          nu *= 2.0;
          if (!(Ft == Ft) && chol_ok) {
            // a non-finite trial value is a rejected step; a non-finite CURRENT value is caught below
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
      if (lane < nv) nonfinite = nonfinite || !(x[lane] == x[lane]) || fabs(x[lane]) > 1e30;
      nonfinite = __any(nonfinite);
      if (nonfinite) status = ST_FALLBACK;  // like optimizer.py:100-102: last_qpos is returned
      if (lane < nv) {
        const double v = nonfinite ? xl[lane] : x[lane];
        if (nonfinite) x[lane] = v;
        kp.qout[it * ld + tb.var_api[lane]] = (float)v;
        if (kp.qout64) kp.qout64[it * ld + tb.var_api[lane]] = v;
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
}

}  // namespace dexr
