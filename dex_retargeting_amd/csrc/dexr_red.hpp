// dexr_red.hpp -- solve kernel in REDUCED VARIABLES: one lane per frame, Hessian of the n_var optimised variables
// in registers, kinematics of the (up to 32) joints in LDS.
//
// Why: with mimic joints the number of joints a component moves (Ability 10, Inspire 12, Schunk SVH 20, + 6 dummy
// joints for position models) is two to three times the number of VARIABLES it optimises (6 / 6 / 9): the reference folds
// the mimic columns into the source joint's column (kinematics_adaptor.py:107-113).  The joint-space kernels
// (dexr_kernel.hpp, dexr_big.hpp) assemble and factor an n_joint x n_joint Hessian and fold afterwards -- 300 entries for
// SVH, which pushed those models onto the spilling 24-joint register kernel + float64 polish (45 ms per 65 536 frames)
// or the LDS kernel (23 ms).  Here the fold happens where the Jacobian column is formed:
//   * the joint loops (forward kinematics, Jacobian columns) are ROLLED: joint k is a run-time, wave-uniform index; world
//     axes / origins live in LDS ([row][lane], conflict-free) and the tables arrive through scalar loads;
//   * per residual term the columns of the chain joints are accumulated straight into the NV variable columns
//     colv[var[k]] += vmul[k] * col_k, a wave-uniform switch on var[k] (registers need static indices);
//   * the second-order (Newton) term sum_{j in fam a, k in fam b} m_j m_k f.(a_min x col_max) is accumulated in the same
//     sweep from running per-variable axis sums A[v] = sum_{j <= k, j in fam v} m_j a_j;
//   * value, residuals and kinematics in float64 (no polish launch needed, see dexr_big.hpp), gradient / Hessian /
//     Cholesky in float32, persistent lanes with a per-component frame queue as in dexr_big.hpp.
// The instruction stream is a few KB per stage (one copy of each rolled loop) instead of the 100+ KB of the unrolled
// kernels.  NV buckets: 8 and 16.
#pragma once

#include "dexr_big.hpp"  // sincos_f64, BIG_NSLOT

#ifndef DEXR_RED_MINW8
#define DEXR_RED_MINW8 1  // waves per SIMD the NV = 8 instantiation must leave room for (2: 120 B of scratch per lane)
#endif

namespace dexr {

// blockDim.x = 64 (one wave per block); dynamic LDS = 64 * (4 * 6 * red_nj + 8 * 3 * lds_frames) bytes:
// float32 axes + origins of red_nj joints, float64 positions of lds_frames frames, [row][lane].
template <int NV>
__global__ void __launch_bounds__(64, (NV <= 8 ? DEXR_RED_MINW8 : 1)) dexr_red_kernel(const KernelParams kp, const dexr_comp_table* __restrict__ comps) {
  constexpr int NH = NV * (NV + 1) / 2;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = blockIdx.x;
  const int comp = (int)(wave_global % kp.n_comp);
  const int64_t tile = wave_global / kp.n_comp;
  int64_t item = 0;
  bool active = false;

  const int NJL = kp.red_nj;  // joints per component the LDS rows were sized for
  // Joint axes and origins in float32 (origins relative to c0, the float64 origin of the first revolute joint: the hand
  // may sit 0.5 m from the world origin, its lever arms are centimetres); frame positions in float64.
  float* AXl = reinterpret_cast<float*>(lds_raw) + lane;                                 // axis of joint k: AXl[(3k+i)*64]
  float* OGl = AXl + (size_t)3 * NJL * 64;                                                // origin - c0
  double* Pl = reinterpret_cast<double*>(lds_raw + (size_t)6 * NJL * 64 * 4) + lane;      // frame f at Pl[(3f+i)*64]
  constexpr auto hidx = [](int r, int c) constexpr { return r * (r + 1) / 2 + c; };

  const dexr_comp_table& tb = comps[comp];
  const int nj = tb.n_joint, nt = tb.n_term, nv = tb.n_var;
  const float delta = kp.norm_delta;
  const int64_t nB = kp.bucket ? (int64_t)kp.bucket[1] : kp.B;
  const int64_t pbase = kp.bucket ? (int64_t)kp.bucket[0] : 0;
  auto row_of = [&](int64_t it) -> int64_t { return kp.perm ? (int64_t)kp.perm[pbase + it] : it; };
  const int ld = kp.ld;
  const bool seq = kp.T > 0;
  int64_t lrow = 0, irow = 0;
  int t_seq = 0;
  const float* lastp = kp.last;

  uint32_t revmask = 0, movmask = 0;  // revolute joints / joints that move with a variable (wave-uniform bit masks)
#pragma clang loop unroll(disable)
  for (int k = 0; k < nj; ++k) {
    if (tb.jtype[k] == DEXR_JOINT_REVOLUTE) revmask |= 1u << k;
    if (tb.var[k] >= 0) movmask |= 1u << k;
  }
  // per-variable constants (static index -> registers / SGPRs)
  float lo_v[NV], hi_v[NV];
  int api_v[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int jv = v < nv ? tb.var_joint[v] : 0;
    lo_v[v] = v < nv ? tb.lo[jv] : 0.f;
    hi_v[v] = v < nv ? tb.hi[jv] : 0.f;
    api_v[v] = v < nv ? tb.api[jv] : 0;
  }

  // ---- per-lane register state ---------------------------------------------------------------------------------
  float x[NV], xo[NV], g[NV], d[NV], H[NH];

  auto ref_row = [&](int row, float (&rv)[3]) {
    if (kp.kpts) {
      const float* a = kp.kpts + (irow * kp.n_kp + kp.h_task[row]) * 3;
      const int o = kp.h_origin[row];
      if (o >= 0) {
        const float* b = kp.kpts + (irow * kp.n_kp + o) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = a[i] - b[i];
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) rv[i] = a[i];
      }
    } else {
      const float* r = kp.ref + (irow * kp.n_ref + row) * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) rv[i] = r[i];
    }
  };
  // regularisation target of variable v (static v): `last` of the item, or the previous frame's answer in sequence mode
  auto xl = [&](int v) -> float {
    float val;
    if (seq && t_seq > 0)
      val = __hip_atomic_load(const_cast<float*>(lastp) + api_v[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      val = lastp[api_v[v]];
    return seq ? fminf(fmaxf(val, lo_v[v] + kp.clip_eps), hi_v[v] - kp.clip_eps) : val;
  };

  uint32_t nst = 0;
  const bool dexpilot = kp.kind == DEXR_KIND_DEXPILOT;
  const int F_ = kp.num_fingers, n_pair = F_ * (F_ - 1) / 2, len_s1 = F_ - 1;
  auto load_frame = [&](int64_t it, int t) {
    item = it;
    t_seq = t;
    lrow = row_of(it);
    irow = seq ? (int64_t)t * kp.seq_stride + lrow : lrow;
    lastp = (seq && t > 0) ? kp.qout + (irow - kp.seq_stride) * ld : kp.last + lrow * ld;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      x[v] = 0;
      if (v < nv) {
        const float s = (kp.x0 && !(seq && t > 0)) ? kp.x0[lrow * ld + api_v[v]] : xl(v);
        x[v] = fminf(fmaxf(s, lo_v[v]), hi_v[v]);
      }
    }
    if (dexpilot) {
      const uint32_t st = (seq && t > 0) ? nst : (kp.state ? kp.state[lrow] : 0u);
      nst = 0;
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
      for (int a = 0; a < F_ - 2; ++a)
        for (int b2 = a + 1; b2 < F_ - 1; ++b2) {
          float rv[3];
          ref_row(idx, rv);
          const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
          const bool b = ((nst >> b2) & 1u) && ((nst >> a) & 1u) && (dist <= 0.03f);
          nst |= (b ? 1u : 0u) << idx;
          ++idx;
        }
    }
  };
  auto term_target = [&](int row, float (&tv)[3], float& wt) {
    float rv[3];
    ref_row(row, rv);
    wt = 1.f;
    if (dexpilot) {
      if (row < n_pair) {
        if ((nst >> row) & 1u) {
          const float dist = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
          const float eta = row < len_s1 ? kp.eta1 : kp.eta2;
#pragma unroll
          for (int i = 0; i < 3; ++i) tv[i] = (rv[i] / (dist + 1e-6f)) * eta;
          wt = row < len_s1 ? 200.f : 400.f;
          return;
        }
      } else {
        wt = (float)(n_pair + F_);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) tv[i] = rv[i] * kp.scaling;
    } else {
      const float sc = (kp.kind == DEXR_KIND_VECTOR) ? kp.scaling : 1.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) tv[i] = rv[i] * sc;
    }
  };

#pragma clang loop unroll(disable) vectorize(disable)
  for (int f = 0; f < tb.n_base_frame; ++f) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Pl[(f * 3 + i) * 64] = (double)tb.frame_off[f][i];
  }

  // value of variable var (wave-uniform run-time index) from the register array: masked sum, no dynamic indexing
  auto pick_x = [&](int var) -> float {
    float v = 0;
#pragma unroll
    for (int s = 0; s < NV; ++s) v += (s == var ? 1.f : 0.f) * x[s];
    return v;
  };

  // ---- float64 forward kinematics, rolled over the joints -----------------------------------------------------------
  double c0[3] = {0, 0, 0};  // origin of the first revolute joint: reference point of the float32 lever arms
  auto fk = [&]() {
    bool c0_set = false;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    double sR[BIG_NSLOT][9], sp[BIG_NSLOT][3];
#pragma unroll
    for (int s = 0; s < BIG_NSLOT; ++s) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sR[s][i] = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) sp[s][i] = 0;
    }
#pragma clang loop unroll(disable) vectorize(disable)
    for (int k = 0; k < nj; ++k) {
      const int rs = tb.restore[k];
      if (rs == -2) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        p[0] = 0; p[1] = 0; p[2] = 0;
      } else if (rs >= 0) {
#pragma unroll
        for (int s = 0; s < BIG_NSLOT; ++s)
          if (rs == s) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = sR[s][i];
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = sp[s][i];
          }
      }
      const float* Xk = tb.X[k];
#pragma unroll
      for (int i = 0; i < 3; ++i) p[i] += R[3 * i] * (double)Xk[9] + R[3 * i + 1] * (double)Xk[10] + R[3 * i + 2] * (double)Xk[11];
      double Rn[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          Rn[3 * i + j] = R[3 * i] * (double)Xk[j] + R[3 * i + 1] * (double)Xk[3 + j] + R[3 * i + 2] * (double)Xk[6 + j];
      // joint value: variable (own or mimicked, kinematics_adaptor.py:102-105) or caller-supplied fixed value
      // (in float64: a mimic joint's value rounded to float32 would make F a step function of x at the 1e-9 level --
      // enough to reject every Newton step in the last 1e-4 rad of a flat valley)
      const int var = tb.var[k];
      double q;
      if (var >= 0) q = (double)tb.vmul[k] * (double)pick_x(var) + (double)tb.off[k];
      else q = (double)tb.mult[k] * (double)kp.fixed[irow * kp.ldf + tb.src_idx[k]] + (double)tb.off[k];
      const bool rev = (revmask >> k) & 1u;
      if (rev) {
        double s, c;
        sincos_f64(q, &s, &c);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double c0_ = Rn[3 * i], c1_ = Rn[3 * i + 1];
          R[3 * i] = c * c0_ + s * c1_;
          R[3 * i + 1] = c * c1_ - s * c0_;
          R[3 * i + 2] = Rn[3 * i + 2];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] += q * Rn[3 * i + 2];
      }
      if (!c0_set && rev) {
        c0_set = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) c0[i] = p[i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        AXl[(3 * k + i) * 64] = (float)R[3 * i + 2];
        OGl[(3 * k + i) * 64] = (float)(p[i] - c0[i]);
      }
      const int sv = tb.save[k];
      if (sv >= 0) {
#pragma unroll
        for (int s = 0; s < BIG_NSLOT; ++s)
          if (sv == s) {
#pragma unroll
            for (int i = 0; i < 9; ++i) sR[s][i] = R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) sp[s][i] = p[i];
          }
      }
      const int fb = tb.fbeg[k], fe = tb.fend[k];
#pragma clang loop unroll(disable) vectorize(disable)
      for (int f = fb; f < fe; ++f) {
        const double o0 = tb.frame_off[f][0], o1 = tb.frame_off[f][1], o2 = tb.frame_off[f][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) Pl[(f * 3 + i) * 64] = p[i] + R[3 * i] * o0 + R[3 * i + 1] * o1 + R[3 * i + 2] * o2;
      }
    }
  };

  const bool per_coord = kp.kind == DEXR_KIND_POSITION;
  const double beta = (double)kp.huber_delta, ibeta = 1.0 / beta;
  const bool newton = kp.newton != 0;

  // ---- fused value / gradient / Hessian in the reduced variables at the FK state --------------------------------------
  auto assemble = [&]() -> double {
    double Fv = 0;
    double gd_[NV];  // gradient of the data term, accumulated in float64 (forces of ~10 cancel to ~0 with DexPilot's weights)
#pragma unroll
    for (int i = 0; i < NH; ++i) H[i] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) gd_[v] = 0.0;
#pragma clang loop unroll(disable) vectorize(disable)
    for (int t = 0; t < nt; ++t) {
      const int ft = tb.term_task[t], fo = tb.term_origin[t];
      float tv[3], wt;
      term_target(tb.term_ref[t], tv, wt);
      double ptd[3], pod[3] = {0, 0, 0}, rd[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) ptd[i] = Pl[(ft * 3 + i) * 64];
      if (fo >= 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) pod[i] = Pl[(fo * 3 + i) * 64];
      }
      float r[3], fvec[3], hw[3], kap = 0.f;
      double fd[3];  // force dF/dr in float64 for the gradient
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        rd[i] = ptd[i] - pod[i] - (double)tv[i];
        r[i] = (float)rd[i];
      }
      const double w = (double)kp.inv_norm * (double)wt;
      if (per_coord) {  // SmoothL1 per coordinate (optimizer.py:130,166)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double ee = rd[i], ae = fabs(ee);
          const bool quad = ae < beta;
          Fv += w * (quad ? 0.5 * ee * ee * ibeta : ae - 0.5 * beta);
          fd[i] = w * (quad ? ee * ibeta : (ee > 0 ? 1.0 : -1.0));
          fvec[i] = (float)fd[i];
          hw[i] = (float)(w * (quad ? ibeta : (newton ? 0.0 : 1.0 / ae)));  // exact curvature in Newton mode
        }
      } else {  // SmoothL1 of the vector norm (optimizer.py:272-273, 534-541)
        const double d2 = rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2];
        const double dd = sqrt(d2);
        const bool quad = dd < beta;
        Fv += w * (quad ? 0.5 * d2 * ibeta : dd - 0.5 * beta);
        const double id = quad ? ibeta : 1.0 / dd;
        const double psi = w * id;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          fd[i] = psi * rd[i];
          fvec[i] = (float)fd[i];
          hw[i] = (float)psi;
        }
        kap = quad ? 0.f : (float)(psi * id * id);
      }
      // folded Jacobian columns of the term, one per variable (zero where the term does not depend on it)
      float colv[NV][3];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        colv[v][0] = 0.f; colv[v][1] = 0.f; colv[v][2] = 0.f;
      }
      uint32_t vterm = 0;  // variables the term depends on
      // one sweep per kinematic chain of the term: the task frame's (force +f) and the origin frame's (force -f).  A joint
      // that is an ancestor of both is visited twice; its two column parts add up to a x (p_task - p_origin).
#pragma clang loop unroll(disable)
      for (int chain = 0; chain < 2; ++chain) {
        if (chain == 1 && fo < 0) break;
        const int fr = chain == 0 ? ft : fo;
        const float sg = chain == 0 ? 1.f : -1.f;
        const float pf0 = (float)((chain == 0 ? ptd[0] : pod[0]) - c0[0]), pf1 = (float)((chain == 0 ? ptd[1] : pod[1]) - c0[1]),
                    pf2 = (float)((chain == 0 ? ptd[2] : pod[2]) - c0[2]);
        uint32_t todo = tb.frame_anc[fr] & movmask;
        // running sums A[v] = sum over the chain joints j visited so far that move with variable v of vmul[j] * axis_j
        // (revolute joints only: a prismatic joint has no second derivative of its own)
        float A[NV][3];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          A[v][0] = 0.f; A[v][1] = 0.f; A[v][2] = 0.f;
        }
        uint32_t vseen = 0;
        while (todo != 0u) {
          const int k = __builtin_ctz(todo);  // wave-uniform: ancestors come in increasing joint order
          todo &= todo - 1u;
          const float a0 = AXl[(3 * k) * 64], a1 = AXl[(3 * k + 1) * 64], a2 = AXl[(3 * k + 2) * 64];
          const bool rev = (revmask >> k) & 1u;
          float c0_, c1_, c2_;
          if (rev) {
            const float v0 = pf0 - OGl[(3 * k) * 64], v1 = pf1 - OGl[(3 * k + 1) * 64], v2 = pf2 - OGl[(3 * k + 2) * 64];
            c0_ = sg * (a1 * v2 - a2 * v1);
            c1_ = sg * (a2 * v0 - a0 * v2);
            c2_ = sg * (a0 * v1 - a1 * v0);
          } else {
            c0_ = sg * a0; c1_ = sg * a1; c2_ = sg * a2;
          }
          const int var = tb.var[k];
          const float m = tb.vmul[k];
          // e = col x f: d2p/dq_j dq_k . f = a_j . (col_k x f) for j an ancestor-or-self of k on this chain
          const float e0 = c1_ * fvec[2] - c2_ * fvec[1];
          const float e1 = c2_ * fvec[0] - c0_ * fvec[2];
          const float e2 = c0_ * fvec[1] - c1_ * fvec[0];
          const float self2 = rev ? m * m * (a0 * e0 + a1 * e1 + a2 * e2) : 0.f;
#pragma unroll
          for (int wv = 0; wv < NV; ++wv) {
            if (var == wv) {  // wave-uniform: exactly one body runs
              if (newton) {
#pragma unroll
                for (int vv = 0; vv < NV; ++vv) {
                  if ((vseen >> vv) & 1u) {
                    const float h = m * (A[vv][0] * e0 + A[vv][1] * e1 + A[vv][2] * e2);
                    if (vv == wv) H[hidx(wv, wv)] += 2.f * h;  // both orders of a pair inside one family
                    else H[vv > wv ? hidx(vv, wv) : hidx(wv, vv)] += h;
                  }
                }
                H[hidx(wv, wv)] += self2;
                if (rev) {
                  A[wv][0] += m * a0; A[wv][1] += m * a1; A[wv][2] += m * a2;
                }
              }
              colv[wv][0] += m * c0_; colv[wv][1] += m * c1_; colv[wv][2] += m * c2_;
            }
          }
          vseen |= 1u << var;
        }
        vterm |= vseen;
      }
      // gradient and Gauss-Newton part (+ exact SmoothL1-of-norm curvature) over the term's variables
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if ((vterm >> v) & 1u) {
          gd_[v] += (double)colv[v][0] * fd[0] + (double)colv[v][1] * fd[1] + (double)colv[v][2] * fd[2];
          const float ku = kap * (colv[v][0] * r[0] + colv[v][1] * r[1] + colv[v][2] * r[2]);
          const float cw0 = hw[0] * colv[v][0] - ku * r[0], cw1 = hw[1] * colv[v][1] - ku * r[1],
                      cw2 = hw[2] * colv[v][2] - ku * r[2];
#pragma unroll
          for (int c = 0; c <= v; ++c)
            if ((vterm >> c) & 1u) H[hidx(v, c)] += cw0 * colv[c][0] + cw1 * colv[c][1] + cw2 * colv[c][2];
        }
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      g[v] = 0.f;
      if (v < nv) {
        const double dx = (double)x[v] - (double)xl(v);
        Fv += (double)delta * dx * dx;
        g[v] = (float)(gd_[v] + 2.0 * (double)delta * dx);  // regulariser's gradient added here, in float64
      }
    }
    return Fv;
  };

  // ---- in-place register Cholesky + solve (H + mask / damping) dvec = -g ----------------------------------------------
  float hdmean = 0.f;
  auto factor_and_solve = [&](uint32_t freemask, float lam) -> bool {
    bool ok = true;
    float hds = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const bool fr = (freemask >> r) & 1u;
#pragma unroll
      for (int c = 0; c < r; ++c) {
        const bool fc = (freemask >> c) & 1u;
        if (!(fr && fc)) H[hidx(r, c)] = 0.f;
      }
      hds += fr ? H[hidx(r, r)] : 0.f;
      H[hidx(r, r)] = fr ? H[hidx(r, r)] + 2.f * delta + lam : 1.f;
    }
    const int hdn = __popc(freemask);
    hdmean = hds / (float)(hdn > 0 ? hdn : 1);
    float inv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float dj = H[hidx(j, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) dj -= H[hidx(j, k)] * H[hidx(j, k)];
      // modified Cholesky (see LaneSolver::chol_solve): a non-positive pivot -- indefinite Newton model, frequent with
      // mimic families -- is reflected instead of failing the pass; `ok` then only says "unmodified Newton step"
      if (!(dj > 1e-6f * (2.f * delta + lam))) {
        ok = false;
        dj = fmaxf(fabsf(dj), 2.f * delta + lam);
      }
      const float iv = __frsqrt_rn(dj);
      inv[j] = iv;
      H[hidx(j, j)] = dj * iv;
#pragma unroll
      for (int i = j + 1; i < NV; ++i) {
        float s = H[hidx(i, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= H[hidx(i, k)] * H[hidx(j, k)];
        H[hidx(i, j)] = s * iv;
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = ((freemask >> i) & 1u) ? -g[i] : 0.f;
#pragma unroll
      for (int k = 0; k < i; ++k) s -= H[hidx(i, k)] * d[k];
      d[i] = s * inv[i];
    }
#pragma unroll
    for (int i = NV - 1; i >= 0; --i) {
      float s = d[i];
#pragma unroll
      for (int k = i + 1; k < NV; ++k) s -= H[hidx(k, i)] * d[k];
      d[i] = s * inv[i];
    }
    return ok;
  };

  // ---- projected Levenberg-Marquardt / Newton: the loop of dexr_big.hpp ----------------------------------------------
  const uint32_t varmask = nv >= 32 ? 0xffffffffu : ((1u << nv) - 1u);
  float lam = kp.lam0, nu = 2.f, sprev = 1e30f;
  bool done = true, pending = false;
  int status = ST_MAXITER, my_iters = 0, blind = 0, my_pass = 0;
  double F = 0;
  float smax = 0, pred = 0;
  bool ok = true;
  const int max_pass = 2 * kp.max_iter + 2;
  const bool in_static = tile * 64 < (int64_t)kp.q0 && tile * 64 < nB;
  unsigned pool_next = in_static ? (unsigned)(tile * 64) : 0u;
  unsigned pool_end = in_static ? (unsigned)((tile * 64 + 64 < nB) ? tile * 64 + 64 : nB) : 0u;
  bool dry = false;
  unsigned* queue = kp.queue + comp;
  for (;;) {
    const unsigned long long want = __ballot(!active);
    if (want != 0ull) {
      if (pool_next >= pool_end && !dry) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(queue, 64u);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + kp.q0;
        if ((int64_t)base >= nB) {
          dry = true;
        } else {
          pool_next = base;
          pool_end = (unsigned)(((int64_t)base + 64 < nB) ? base + 64 : nB);
        }
      }
      if (pool_next < pool_end) {
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
        const unsigned cand = pool_next + rank;
        const bool got = !active && cand < pool_end;
        pool_next += (unsigned)__popcll(__ballot(got));
        if (got) {
          load_frame((int64_t)cand, 0);
          active = true;
          done = false;
          pending = false;
          lam = kp.lam0;
          nu = 2.f;
          sprev = 1e30f;
          status = ST_MAXITER;
          my_iters = 0;
          blind = 0;
          my_pass = 0;
          F = 0;
          smax = 0;
          pred = 0;
          ok = true;
        }
      }
    }
    if (!__any(active)) {
      if (dry && pool_next >= pool_end) break;
      continue;
    }
    fk();
    const double Fe = assemble();
    bool rebuild = false;
    if (!done) {
      if (!pending) {
        F = Fe;
      } else {
        const double noise = (double)kp.floor_scale * fabs(F);
        const bool finite = (Fe == Fe) && (smax == smax) && (fabs(Fe) < 1e30);
        const bool below_floor = ok && finite && ((double)pred <= noise) && (smax < 1e-2f);
        const bool accept = finite && ((Fe <= F) || below_floor);  // (a modified-Cholesky step is judged by the decrease)
        ++my_iters;
        pending = false;
        if (accept) {
          const float rho = (float)((F - Fe) / fmax((double)pred, 1e-30));
          const float tt = 2.f * rho - 1.f;
          float shrink = below_floor ? (1.f / 3.f) : fmaxf(1.f / 3.f, 1.f - tt * tt * tt);
          if (kp.lam_fastdec > 0 && rho > 0.9f) shrink = kp.lam_fastdec;
          lam = fmaxf(lam * shrink, 1e-9f);
          nu = 2.f;
          F = Fe;
          const bool stalled = below_floor && blind >= kp.stall_from && smax > kp.stall_ratio * sprev && smax < kp.stall_cap * kp.tol;
          blind = below_floor ? blind + 1 : 0;
          sprev = smax;
          // A step below tol only means convergence when the damping is not what made it small: with lambda far above
          // the weakest curvature the model can have (the regulariser's 2 delta) a step of 1e-8 says nothing about the
          // distance to the minimiser (mimic DexPilot models: frames stopped 1e-4..8e-4 rad short after a rejected step
          // had raised lambda).  Such a step shrinks lambda tenfold instead and the iteration goes on.
          const float lam_ok = fmaxf(2.f * delta, 10.f * kp.lam0);
          if ((smax < kp.tol && lam <= lam_ok) || stalled || blind >= kp.max_blind) {
            done = true;
            status = ST_CONVERGED;
          } else if (smax < kp.tol) {
            lam = fmaxf(0.1f * lam, 0.5f * lam_ok);
          }
        } else {
          lam = fmaxf(lam, 1e-6f) * nu;
          if (kp.lam_jump > 0) lam = fmaxf(lam, kp.lam_jump * hdmean);
          nu *= 2.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) x[v] = xo[v];
          if (lam > 1e10f) {
            done = true;
            status = finite ? ST_CONVERGED : ST_FALLBACK;
          }
          // a step shorter than tol that does not decrease F: the decrease along the (damped) descent direction is below
          // the resolution of F -- converged at the rounding floor (and no livelock between tiny accepted steps that
          // shrink lambda and rounding-level rejections that raise it again)
          if (finite && smax < kp.tol) {
            done = true;
            status = ST_CONVERGED;
          }
          rebuild = true;
        }
        if (!done && my_iters >= kp.max_iter) done = true;
      }
    }
    uint32_t freemask = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if ((varmask >> v) & 1u) {  // (g already holds data term + regulariser, see assemble)
        const bool act = (x[v] <= lo_v[v] && g[v] > 0) || (x[v] >= hi_v[v] && g[v] < 0);
        if (!act) freemask |= 1u << v;
      } else {
        g[v] = 0;
      }
    }
    const bool okf = factor_and_solve(freemask, lam);
    const bool stepping = !done && !rebuild;
    if (stepping) {
      ok = okf;
      smax = 0;
      pred = 0;
    }
    float dmax = 0.f, gd = 0.f, dd = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if ((freemask >> v) & 1u) {
        dmax = fmaxf(dmax, fabsf(d[v]));
        gd -= g[v] * d[v];
        dd += d[v] * d[v];
      }
    // trust radius: a step whose largest component exceeds step_cap is scaled down to it; a step from a MODIFIED
    // factorisation (negative curvature along the way: the quadratic model has no minimiser in that direction) is scaled
    // UP towards it (at most 8 x: an unbounded stretch overshoots on the SVH hand's saddles) -- near a degenerate saddle the reflected-pivot step is ~g / (2 delta) with g -> 0 and would crawl for
    // dozens of passes (one such frame in 65 536 set the duration of a whole launch)
    const float alpha = (kp.step_cap > 0 && (dmax > kp.step_cap || (!okf && dmax > 0.f))) ? fminf(kp.step_cap / dmax, 8.f) : 1.f;
    if (stepping) pred = alpha * (1.f - 0.5f * alpha) * gd + 0.5f * alpha * alpha * lam * dd;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (stepping) {
        xo[v] = x[v];
        if ((freemask >> v) & 1u) {
          const float xt = fminf(fmaxf(x[v] + alpha * d[v], lo_v[v]), hi_v[v]);
          smax = fmaxf(smax, fabsf(xt - x[v]));
          x[v] = xt;
        }
      }
    }
    pending = stepping;
    if (stepping && okf && smax < kp.blind_tol && lam <= kp.lam0) {
      ++my_iters;
      pending = false;
      done = true;
      status = ST_CONVERGED;
    }
    if (active && !done && ++my_pass >= max_pass) done = true;

    if (active && done) {
      if (pending) {
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v] = xo[v];
        pending = false;
      }
      bool bad = false;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if ((varmask >> v) & 1u) bad = bad || !(x[v] == x[v]);
      if (bad) status = ST_FALLBACK;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if ((varmask >> v) & 1u) {
          const float val = bad ? xl(v) : x[v];
          kp.qout[irow * ld + api_v[v]] = val;
          if (kp.qout64) kp.qout64[irow * ld + api_v[v]] = (double)val;
        }
      }
      if (kp.status) atomicMax(&kp.status[irow], status);
      if (kp.iters) atomicMax(&kp.iters[irow], my_iters);
      if (kp.fval) atomicAdd(&kp.fval[irow], (float)F);
      if (seq && t_seq + 1 < kp.T) {
        load_frame(item, t_seq + 1);
        done = false;
        pending = false;
        lam = kp.lam0;
        nu = 2.f;
        sprev = 1e30f;
        status = ST_MAXITER;
        my_iters = 0;
        blind = 0;
        my_pass = 0;
        F = 0;
        smax = 0;
        pred = 0;
        ok = true;
      } else {
        if (dexpilot && kp.state && comp == 0) kp.state[lrow] = nst;
        active = false;
      }
    }
  }
}

}  // namespace dexr
