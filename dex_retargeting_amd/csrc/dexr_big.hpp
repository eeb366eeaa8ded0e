// dexr_big.hpp -- solve kernel for LARGE components (16..32 joints per lane): DexPilot hands, hands with a shared
// wrist or 6 free joints.
//
// Why a second kernel: a dense 24 x 24 Hessian is 300 floats per lane; together with the chain's axes/origins and the
// Jacobian columns that is > 512 live values, and the register-resident kernel (dexr_kernel.hpp) spills 1.4 KB per
// lane to scratch memory (15 ms per 65 536 Shadow-DexPilot frames).  Here
//   * the Hessian / Cholesky factor lives in the lane's LDS column ([entry][lane]: conflict-free ds ops, one
//     ds_add_f32 per accumulated entry), the factorisation is a rolled right-looking Cholesky whose pivot column is
//     cached in registers, and its work scales with the component's real joint count n, not the bucket size;
//   * forward kinematics, frame positions, residuals and the objective value are computed in FLOAT64 (the residual of a
//     vector term is a difference of positions ~0.2 m that nearly cancels; in float32 that rounding puts a ~1e-4 rad
//     floor under DexPilot/position solves), while Jacobian columns, gradient, Hessian and the linear solve stay in
//     FLOAT32 -- so no float64 polish launch is needed for these models;
//   * per-term targets / DexPilot weights are recomputed from ref_value (L1/L2 hits) instead of occupying LDS.
// MI355X: 160 KB LDS/CU, so 2 waves/CU at n = 22 (LEAP/Allegro + free joints), 1 wave/CU at n >= 24.
#pragma once

#include "dexr_kernel.hpp"
#include "dexr_math.hpp"  // sincos_f64

namespace dexr {

constexpr int BIG_NSLOT = 2;  // saved transforms kept in registers (float64); deeper forks use the generic kernel

// blockDim.x = 64 (one wave per block); dynamic LDS = 64 * (4 * nh_rows + 8 * 3 * lds_frames) bytes where
// nh_rows = n_max (n_max + 1) / 2 for the model's largest component.
template <int NMAX>
__global__ void __launch_bounds__(64) dexr_big_kernel(const KernelParams kp, const dexr_comp_table* __restrict__ comps) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = blockIdx.x;
  const int comp = (int)(wave_global % kp.n_comp);
  const int64_t tile = wave_global / kp.n_comp;  // this wave's index among the waves of its component
  // PERSISTENT LANES (as in dexr_quad.hpp): a lane that finishes its frame stores it and takes the next one -- wave w
  // starts with the static tile [64 w, 64 w + 64), frames from kp.q0 on are handed out by the per-component queue.
  int64_t item = 0;     // work item (frame, or sequence) this lane is working on
  bool active = false;  // the lane holds a frame

  float* Hl = reinterpret_cast<float*>(lds_raw) + lane;                                 // H(r,c) at Hl[hidx(r,c)*64]
  double* Pl = reinterpret_cast<double*>(lds_raw + (size_t)kp.big_nh_rows * 64 * 4) + lane;  // frame f at Pl[(3f+i)*64]
  auto hidx = [](int r, int c) { return r * (r + 1) / 2 + c; };

  const dexr_comp_table& tb = comps[comp];
  const int nj = tb.n_joint, nt = tb.n_term;
  const float delta = kp.norm_delta;
  // number of work items and their rows: fleet buckets are sized and listed on the device (see KernelParams)
  const int64_t nB = kp.bucket ? (int64_t)kp.bucket[1] : kp.B;
  const int64_t pbase = kp.bucket ? (int64_t)kp.bucket[0] : 0;
  auto row_of = [&](int64_t it) -> int64_t { return kp.perm ? (int64_t)kp.perm[pbase + it] : it; };
  const int ld = kp.ld;
  const bool seq = kp.T > 0;  // sequence mode: a work item is a sequence of kp.T frames solved in order by this lane
  int64_t lrow = 0, irow = 0;  // row of the item's `last` / `state`; row of the current frame's inputs and outputs
  int t_seq = 0;               // frame of the sequence being solved
  const float* lastp = kp.last;

  uint32_t vmask = 0, optmask = 0;
  uint32_t revmask = 0;  // revolute joints: a wave-uniform bit mask, so the loops below test a bit instead of loading jtype
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k < nj) {
      if (tb.jtype[k] == DEXR_JOINT_REVOLUTE) revmask |= 1u << k;
      const int sk = tb.src_kind[k];
      if (sk == DEXR_SRC_OPT) { vmask |= 1u << k; optmask |= 1u << k; }
      else if (sk == DEXR_SRC_MIMIC) vmask |= 1u << k;
    }
  }

  // ---- per-lane register state -----------------------------------------------------------------------------
  float x[NMAX], xo[NMAX], g[NMAX], d[NMAX];
  float ax[NMAX][3], og[NMAX][3];

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
  // regularisation target = start point of the frame (L1/L2 hit): `last` of the item, or -- sequence mode, frame
  // t > 0 -- the previous frame's raw solution, which this lane has just stored to qout; clipped to the joint limits
  // in sequence mode (seq_retarget.py:118-120)
  auto xl = [&](int k) -> float {
    float v;
    if (seq && t_seq > 0)  // written by this wave a moment ago: read it coherently (bypassing the vector L1)
      v = __hip_atomic_load(const_cast<float*>(lastp) + tb.api[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      v = lastp[tb.api[k]];
    return seq ? fminf(fmaxf(v, tb.lo[k] + kp.clip_eps), tb.hi[k] - kp.clip_eps) : v;
  };

  // DexPilot projection bits (optimizer.py:466-476) of the current frame
  uint32_t nst = 0;
  const bool dexpilot = kp.kind == DEXR_KIND_DEXPILOT;
  const int F_ = kp.num_fingers, n_pair = F_ * (F_ - 1) / 2, len_s1 = F_ - 1;
  // ---- load a frame -------------------------------------------------------------------------------------------
  // it: work item; t: frame of the sequence (0 unless sequence mode)
  auto load_frame = [&](int64_t it, int t) {
  item = it;
  t_seq = t;
  lrow = row_of(it);
  irow = seq ? (int64_t)t * kp.seq_stride + lrow : lrow;
  lastp = (seq && t > 0) ? kp.qout + (irow - kp.seq_stride) * ld : kp.last + lrow * ld;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    x[k] = 0;
    if (k < nj) {
      const int sk = tb.src_kind[k];
      if (sk == DEXR_SRC_OPT) {
        const float v = (kp.x0 && !(seq && t_seq > 0)) ? kp.x0[lrow * ld + tb.api[k]] : xl(k);
        x[k] = fminf(fmaxf(v, tb.lo[k]), tb.hi[k]);
      } else if (sk == DEXR_SRC_FIXED) {
        x[k] = tb.mult[k] * kp.fixed[irow * kp.ldf + tb.src_idx[k]] + tb.off[k];
      }
    }
  }
  if (dexpilot) {
    const uint32_t st = (seq && t_seq > 0) ? nst : (kp.state ? kp.state[lrow] : 0u);  // carried bits in sequence mode
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
  // target vector and weight of one term (optimizer.py:246, 479-507), recomputed on demand
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

  // base frames never move
#pragma clang loop unroll(disable) vectorize(disable)
  for (int f = 0; f < tb.n_base_frame; ++f) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Pl[(f * 3 + i) * 64] = (double)tb.frame_off[f][i];
  }

  // ---- float64 forward kinematics ------------------------------------------------------------------------------
  // Lever arms (frame position - joint origin) enter the float32 Jacobian.  With free joints the whole hand sits up to
  // ~0.5 m from the world origin, where float32 resolves 3e-8 m -- 1e-6 of a 3 cm lever arm, which biases the gradient
  // enough to move the fixed point by 1e-5..1e-4 rad on weakly determined joints.  Both operands are therefore taken
  // relative to c0, the float64 origin of the first revolute joint (the hand's base), before the cast.
  double c0[3] = {0, 0, 0};
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
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if (k < nj) {
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
        double q = (double)x[k];
        if (tb.src_kind[k] == DEXR_SRC_MIMIC) {  // kinematics_adaptor.py:102-105
          const int si = tb.src_idx[k];
          float v = 0;
#pragma unroll
          for (int s = 0; s < NMAX; ++s) v += (s == si ? 1.f : 0.f) * x[s];
          // in float64: rounded to float32 the mimic joint's value would make F a step function of x at the 1e-9
          // level, enough to reject every Newton step in the last 1e-4 rad of a flat valley
          q = (double)tb.mult[k] * (double)v + (double)tb.off[k];
          x[k] = (float)q;
        }
        if ((revmask >> k) & 1u) {
          double s, c;
          sincos_f64(q, &s, &c);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double c0 = Rn[3 * i], c1 = Rn[3 * i + 1];
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
        if (!c0_set && ((revmask >> k) & 1u)) {
          c0_set = true;
#pragma unroll
          for (int i = 0; i < 3; ++i) c0[i] = p[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          ax[k][i] = (float)R[3 * i + 2];
          og[k][i] = (float)(p[i] - c0[i]);
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
    }
  };

  const bool per_coord = kp.kind == DEXR_KIND_POSITION;
  const double beta = (double)kp.huber_delta, ibeta = 1.0 / beta;

  // residual of term t in float64, plus (float32) force, curvature weights and the float32 copies the Jacobian needs
  struct TermEval { double val; float fvec[3], hw[3], kap, r[3], pt[3], po[3]; };
  auto eval_term = [&](int t, TermEval& e) {
    const int ft = tb.term_task[t], fo = tb.term_origin[t];
    float tv[3], wt;
    term_target(tb.term_ref[t], tv, wt);
    double pt[3], po[3] = {0, 0, 0}, r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pt[i] = Pl[(ft * 3 + i) * 64];
    if (fo >= 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) po[i] = Pl[(fo * 3 + i) * 64];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      r[i] = pt[i] - po[i] - (double)tv[i];
      e.r[i] = (float)r[i];
      e.pt[i] = (float)(pt[i] - c0[i]);
      e.po[i] = (float)(po[i] - c0[i]);
    }
    const double w = (double)kp.inv_norm * (double)wt;
    e.kap = 0;
    if (per_coord) {  // SmoothL1 per coordinate (optimizer.py:130,166)
      double v = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double ee = r[i], ae = fabs(ee);
        const bool quad = ae < beta;
        v += w * (quad ? 0.5 * ee * ee * ibeta : ae - 0.5 * beta);
        e.fvec[i] = (float)(w * (quad ? ee * ibeta : (ee > 0 ? 1.0 : -1.0)));
        e.hw[i] = (float)(w * (quad ? ibeta : (kp.newton != 0 ? 0.0 : 1.0 / ae)));  // exact curvature in Newton mode
      }
      e.val = v;
    } else {  // SmoothL1 of the vector norm (optimizer.py:272-273, 534-541)
      const double d2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      const double dd = sqrt(d2);
      const bool quad = dd < beta;
      e.val = w * (quad ? 0.5 * d2 * ibeta : dd - 0.5 * beta);
      const double id = quad ? ibeta : 1.0 / dd;
      const double psi = w * id;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        e.fvec[i] = (float)(psi * r[i]);
        e.hw[i] = (float)psi;
      }
      e.kap = quad ? 0.f : (float)(psi * id * id);
    }
  };

  const int nh = nj * (nj + 1) / 2;
  // value F (returned), gradient g (registers) and Hessian H (LDS) of the data term at the FK state, Newton term
  // included; the regulariser's value is added here, its gradient / curvature by the caller.
  auto assemble = [&]() -> double {
    double Fv = 0;
#pragma clang loop unroll(disable) vectorize(disable)
    for (int i = 0; i < nh; ++i) Hl[i * 64] = 0.f;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) g[k] = 0;
    const bool newton = kp.newton != 0;
#pragma clang loop unroll(disable) vectorize(disable)
    for (int t = 0; t < nt; ++t) {
      TermEval e;
      eval_term(t, e);
      Fv += e.val;
      const int ft = tb.term_task[t], fo = tb.term_origin[t];
      const uint32_t mt = tb.frame_anc[ft];
      const uint32_t mo = (fo >= 0) ? tb.frame_anc[fo] : 0u;
      const uint32_t mu = (mt | mo) & vmask;
      float col[NMAX][3];
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if ((mu >> k) & 1u) {
          const bool in_t = (mt >> k) & 1u, in_o = (mo >> k) & 1u;
          if ((revmask >> k) & 1u) {
            float v[3] = {0, 0, 0};
            if (in_t) {
#pragma unroll
              for (int i = 0; i < 3; ++i) v[i] += e.pt[i] - og[k][i];
            }
            if (in_o) {
#pragma unroll
              for (int i = 0; i < 3; ++i) v[i] -= e.po[i] - og[k][i];
            }
            col[k][0] = ax[k][1] * v[2] - ax[k][2] * v[1];
            col[k][1] = ax[k][2] * v[0] - ax[k][0] * v[2];
            col[k][2] = ax[k][0] * v[1] - ax[k][1] * v[0];
          } else {
            const float sg = (in_t ? 1.f : 0.f) - (in_o ? 1.f : 0.f);
#pragma unroll
            for (int i = 0; i < 3; ++i) col[k][i] = sg * ax[k][i];
          }
          g[k] += col[k][0] * e.fvec[0] + col[k][1] * e.fvec[1] + col[k][2] * e.fvec[2];
        } else {
          col[k][0] = 0; col[k][1] = 0; col[k][2] = 0;
        }
      }
#pragma unroll
      for (int rr = 0; rr < NMAX; ++rr) {
        if ((mu >> rr) & 1u) {
          const float ku = e.kap * (col[rr][0] * e.r[0] + col[rr][1] * e.r[1] + col[rr][2] * e.r[2]);
          const float cw0 = e.hw[0] * col[rr][0] - ku * e.r[0], cw1 = e.hw[1] * col[rr][1] - ku * e.r[1],
                      cw2 = e.hw[2] * col[rr][2] - ku * e.r[2];
          const float cf0 = col[rr][1] * e.fvec[2] - col[rr][2] * e.fvec[1];
          const float cf1 = col[rr][2] * e.fvec[0] - col[rr][0] * e.fvec[2];
          const float cf2 = col[rr][0] * e.fvec[1] - col[rr][1] * e.fvec[0];
          // row rr of H: read the whole row segment with independent ds_reads (one wait), update, write back.
          // col[cc] is zero for joints outside the term's chain, so no per-entry guard is needed.
          float hrow[NMAX];
#pragma unroll
          for (int cc = 0; cc <= rr; ++cc) hrow[cc] = Hl[(rr * (rr + 1) / 2 + cc) * 64];
#pragma unroll
          for (int cc = 0; cc <= rr; ++cc) {
            float h = cw0 * col[cc][0] + cw1 * col[cc][1] + cw2 * col[cc][2];
            const bool same = (((mt >> cc) & (mt >> rr)) | ((mo >> cc) & (mo >> rr))) & 1u;
            if (newton && same && ((mu >> cc) & 1u) && ((revmask >> cc) & 1u))
              h += ax[cc][0] * cf0 + ax[cc][1] * cf1 + ax[cc][2] * cf2;
            hrow[cc] += h;
          }
#pragma unroll
          for (int cc = 0; cc <= rr; ++cc) Hl[(rr * (rr + 1) / 2 + cc) * 64] = hrow[cc];
        }
      }
    }
    // mimic fold (kinematics_adaptor.py:107-113): x_k = m x_s + b  =>  congruence on H, fold on g
#pragma clang loop unroll(disable) vectorize(disable)
    for (int k = 0; k < nj; ++k) {
      if (tb.src_kind[k] != DEXR_SRC_MIMIC) continue;
      const int s = tb.src_idx[k];
      const float m = tb.mult[k];
      float gk = 0;
#pragma unroll
      for (int j = 0; j < NMAX; ++j)
        if (j == k) { gk = g[j]; g[j] = 0; }
#pragma unroll
      for (int j = 0; j < NMAX; ++j)
        if (j == s) g[j] += m * gk;
      auto H = [&](int r, int c) -> float& { return r >= c ? Hl[hidx(r, c) * 64] : Hl[hidx(c, r) * 64]; };
      const float hkk = H(k, k), hks = H(k, s);
      for (int j = 0; j < nj; ++j) {
        if (j == k || j == s) continue;
        H(j, s) += m * H(j, k);
      }
      H(s, s) += 2.f * m * hks + m * m * hkk;
      for (int j = 0; j < nj; ++j) H(j, k) = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if ((optmask >> k) & 1u) {
        const double dx = (double)x[k] - (double)xl(k);
        Fv += (double)delta * dx * dx;
      }
    return Fv;
  };

  // ---- in-place Cholesky in LDS (rolled, right-looking) + solve (H + mask/damping) dvec = -g ---------------------
  // freemask: lane-varying set of optimised joints not held at a bound.  Returns false if a pivot is not positive.
  float hdmean = 0.f;  // mean diagonal of the free block of the last factored model (scale for the damping jump)
  auto factor_and_solve = [&](uint32_t freemask, float lam) -> bool {
    bool ok = true;
    // reduced, damped system: rows/cols of held or non-variable joints become identity
    float hds = 0.f;
#pragma clang loop unroll(disable) vectorize(disable)
    for (int r = 0; r < nj; ++r) {
      const bool fr = (freemask >> r) & 1u;
      for (int c = 0; c < r; ++c) {
        const bool fc = (freemask >> c) & 1u;
        if (!(fr && fc)) Hl[hidx(r, c) * 64] = 0.f;
      }
      float& hrr = Hl[hidx(r, r) * 64];
      hds += fr ? hrr : 0.f;
      hrr = fr ? hrr + 2.f * delta + lam : 1.f;
    }
    const int hdn = __popc(freemask);
    hdmean = hds / (float)(hdn > 0 ? hdn : 1);
    // Right-looking Cholesky, one column per (runtime) j.  The body is branch-light on purpose: the pivot column is read
    // with 24 independent ds_reads (inactive rows read the pivot itself), scaled in registers and written back
    // unconditionally; the trailing update is issued as ds_add_f32 (no read, hence no LDS round trip on the critical
    // path) -- entries left of the pivot receive +-0.
    float Lj[NMAX];
#pragma clang loop unroll(disable) vectorize(disable)
    for (int j = 0; j < nj; ++j) {
      const int pj = hidx(j, j);
      float cj[NMAX];
#pragma unroll
      for (int i = 0; i < NMAX; ++i) {
        const bool act = (i > j) && (i < nj);
        cj[i] = Hl[(act ? hidx(i, j) : pj) * 64];
      }
      float dj = Hl[pj * 64];
      if (!(dj > 1e-30f)) { ok = false; dj = 1.f; }
      const float iv = __frsqrt_rn(dj);
      const float sq = dj * iv;
#pragma unroll
      for (int i = 0; i < NMAX; ++i) {
        const bool act = (i > j) && (i < nj);
        Lj[i] = act ? cj[i] * iv : 0.f;
        Hl[(act ? hidx(i, j) : pj) * 64] = act ? Lj[i] : sq;
      }
#pragma unroll
      for (int i = 1; i < NMAX; ++i) {
        if (i > j && i < nj) {  // row i: batched read, rank-one update (Lj[k] = 0 for k <= j), batched write
          const float li = Lj[i];
          float hrow[NMAX];
#pragma unroll
          for (int k = 1; k <= i; ++k) hrow[k] = Hl[hidx(i, k) * 64];
#pragma unroll
          for (int k = 1; k <= i; ++k) Hl[hidx(i, k) * 64] = hrow[k] - li * Lj[k];
        }
      }
    }
    // forward: L y = -g ; backward: L^T dvec = y   (static indices into d[], dynamic guards on n)
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      if (i < nj) {
        float s = ((freemask >> i) & 1u) ? -g[i] : 0.f;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= Hl[hidx(i, k) * 64] * d[k];
        d[i] = s * __frcp_rn(Hl[hidx(i, i) * 64]);
      } else {
        d[i] = 0.f;
      }
    }
#pragma unroll
    for (int i = NMAX - 1; i >= 0; --i) {
      if (i < nj) {
        float s = d[i];
#pragma unroll
        for (int k = i + 1; k < NMAX; ++k)
          if (k < nj) s -= Hl[hidx(k, i) * 64] * d[k];
        d[i] = s * __frcp_rn(Hl[hidx(i, i) * 64]);
      }
    }
    return ok;
  };

  // ---- projected Levenberg-Marquardt / Newton ------------------------------------------------------------------
  // One pass of the loop = forward kinematics + fused value/gradient/Hessian at the point under evaluation
  // (x itself, or the pending trial point) + one factorisation.  FK, assembly and factorisation each appear ONCE in
  // the instruction stream (the unrolled bodies are tens of KB; the 64 KB instruction cache is the scarce resource).
  // A rejected trial costs one extra pass (the model at the old x is rebuilt), like the register kernel's re-run.
  float lam = kp.lam0, nu = 2.f, sprev = 1e30f;
  bool done = true, pending = false;  // done: no solve in progress; pending: x holds an untested trial, xo the accepted point
  int status = ST_MAXITER, my_iters = 0, blind = 0, my_pass = 0;
  double F = 0;
  float smax = 0, pred = 0;
  bool ok = true;
  const int max_pass = 2 * kp.max_iter + 2;
  // wave-uniform pool of unassigned frames
  const bool in_static = tile * 64 < (int64_t)kp.q0 && tile * 64 < nB;
  unsigned pool_next = in_static ? (unsigned)(tile * 64) : 0u;
  unsigned pool_end = in_static ? (unsigned)((tile * 64 + 64 < nB) ? tile * 64 + 64 : nB) : 0u;
  bool dry = false;  // the queue is exhausted
  unsigned* queue = kp.queue + comp;
  for (;;) {
    // (0) hand frames to idle lanes
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
    bool rebuild = false;  // this lane rejected its trial: its model must be rebuilt at xo before it can step again
    if (!done) {
      if (!pending) {
        F = Fe;  // model (re)built at the accepted point
      } else {
        const double noise = (double)kp.floor_scale * fabs(F);
        const bool finite = (Fe == Fe) && (smax == smax) && (fabs(Fe) < 1e30);
        const bool below_floor = ok && finite && ((double)pred <= noise) && (smax < 1e-2f);
        const bool accept = ok && finite && ((Fe <= F) || below_floor);
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
          for (int k = 0; k < NMAX; ++k) x[k] = xo[k];
          if (lam > 1e10f) {
            done = true;
            status = finite ? ST_CONVERGED : ST_FALLBACK;
          }
          // a step shorter than tol that does not decrease F: the decrease along the (damped) descent direction is below
          // the resolution of F -- converged at the rounding floor (and no livelock between tiny accepted steps that
          // shrink lambda and rounding-level rejections that raise it again)
          if (ok && finite && smax < kp.tol) {  // (ok: the step came from a valid factorisation)
            done = true;
            status = ST_CONVERGED;
          }
          rebuild = true;  // the model in g/H belongs to the rejected point: rebuilt at xo in the next pass
        }
        if (!done && my_iters >= kp.max_iter) done = true;
      }
    }
    // step from the model at x
    uint32_t freemask = 0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if ((optmask >> k) & 1u) {
        g[k] += 2.f * delta * (x[k] - xl(k));
        const bool act = (x[k] <= tb.lo[k] && g[k] > 0) || (x[k] >= tb.hi[k] && g[k] < 0);
        if (!act) freemask |= 1u << k;
      } else {
        g[k] = 0;
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
    for (int k = 0; k < NMAX; ++k)
      if ((freemask >> k) & 1u) {
        dmax = fmaxf(dmax, fabsf(d[k]));
        gd -= g[k] * d[k];
        dd += d[k] * d[k];
      }
    // trust radius: step scaled to at most step_cap per joint; predicted decrease of the damped model along alpha*d
    const float alpha = (kp.step_cap > 0 && dmax > kp.step_cap) ? kp.step_cap / dmax : 1.f;
    if (stepping) pred = alpha * (1.f - 0.5f * alpha) * gd + 0.5f * alpha * alpha * lam * dd;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if (stepping) {
        xo[k] = x[k];
        if ((freemask >> k) & 1u) {
          const float xt = fminf(fmaxf(x[k] + alpha * d[k], tb.lo[k]), tb.hi[k]);
          smax = fmaxf(smax, fabsf(xt - x[k]));
          x[k] = xt;
        }
      }
    }
    pending = stepping;
    // verified, undamped model and a Newton step below blind_tol: the step is the converged answer to well below the
    // tolerance -- take it and stop instead of spending one more pass on confirming it
    if (stepping && okf && smax < kp.blind_tol && lam <= kp.lam0) {
      ++my_iters;
      pending = false;
      done = true;
      status = ST_CONVERGED;
    }
    // pass budget of this frame exhausted: hand back the accepted point
    if (active && !done && ++my_pass >= max_pass) done = true;

    // (last) retire finished frames
    if (active && done) {
      if (pending) {  // untested trial point
#pragma unroll
        for (int k = 0; k < NMAX; ++k) x[k] = xo[k];
        pending = false;
      }
      bool bad = false;
#pragma unroll
      for (int k = 0; k < NMAX; ++k)
        if ((optmask >> k) & 1u) bad = bad || !(x[k] == x[k]);
      if (bad) status = ST_FALLBACK;
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if ((optmask >> k) & 1u) {
          const float v = bad ? xl(k) : x[k];
          kp.qout[irow * ld + tb.api[k]] = v;
          if (kp.qout64) kp.qout64[irow * ld + tb.api[k]] = (double)v;
        }
      }
      if (kp.status) atomicMax(&kp.status[irow], status);
      if (kp.iters) atomicMax(&kp.iters[irow], my_iters);
      if (kp.fval) atomicAdd(&kp.fval[irow], (float)F);
      if (seq && t_seq + 1 < kp.T) {
        // next frame of this lane's sequence: start point / regularisation target = the row just written
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
