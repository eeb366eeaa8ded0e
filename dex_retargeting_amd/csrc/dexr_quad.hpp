// dexr_quad.hpp -- solve kernel for LARGE DENSE components (9..24 joints: DexPilot hands, hands with a shared wrist):
// FOUR LANES PER FRAME.
//
// A dense 24 x 24 Hessian does not fit one lane's registers (dexr_kernel.hpp spills 1.4 KB/lane to scratch) and an
// LDS-resident Hessian leaves one wave per CU (dexr_big.hpp).  Here a quad of lanes (4i..4i+3 -- DPP quad_perm moves
// data between them at full VALU rate, no LDS, no barrier) shares one frame:
//   * lane p of the quad owns the Hessian rows r = p (mod 4): 84 floats per lane at n = 24 instead of 300;
//   * the Cholesky factorisation is distributed: the pivot and the pivot column travel by quad broadcast
//     (n^2/2 DPP moves), each lane updates only its own rows (n^3/24 FMAs per lane instead of n^3/6);
//   * forward kinematics, residuals, Jacobian columns, gradient and the step vector are REPLICATED in the four lanes
//     (same instructions, same data, no divergence), which costs redundant flops but no communication;
//   * forward kinematics / residuals / objective value run in float64 (see dexr_big.hpp for why), the Jacobian,
//     Hessian and linear solve in float32: no float64 polish launch for these models.
// A wave holds 16 frames, so 65 536 Shadow-DexPilot frames are 4 096 waves (4 per SIMD) instead of 1 024.
#pragma once

#include "dexr_big.hpp"  // sincos_f64

namespace dexr {

template <int Q>
static __device__ __forceinline__ float quad_bcast(float v) {  // value of lane Q of each quad, to the whole quad
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xF, 0xF, true));
}
static __device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
  return v;
}
static __device__ __forceinline__ float sel4(int p, float a0, float a1, float a2, float a3) {
  const float lo = (p & 1) ? a1 : a0, hi = (p & 1) ? a3 : a2;
  return (p & 2) ? hi : lo;
}

// blockDim.x = 256 (4 waves x 16 frames); dynamic LDS per wave = 64 * (8 * 3 * lds_frames + 4 * 3 * NMAX + 4 * NMAX)
// bytes: float64 frame positions + float32 joint origins + the accepted point, [row][lane] (the 4 lanes of a quad
// hold copies).  Per-term targets
// and DexPilot weights are recomputed from ref_value on demand (L1/L2 hits) to keep 4 waves per CU within 160 KB.
#ifndef DEXR_QUAD_MINW
#define DEXR_QUAD_MINW 1  // waves per SIMD the register allocation must leave room for
#endif
template <int NMAX>
__global__ void __launch_bounds__(256, DEXR_QUAD_MINW) dexr_quad_kernel(const KernelParams kp, const dexr_comp_table* __restrict__ comps) {
  static_assert(NMAX % 4 == 0, "bucket must be a multiple of the quad width");
  constexpr int NR = NMAX / 4;             // Hessian rows per lane
  constexpr int NHQ = 2 * NR * (NR + 1);   // local row i keeps columns 0..4i+3
  auto hq = [](int i) { return 2 * i * (i + 1); };

  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int p = lane & 3;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves_per_block = blockDim.x >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * waves_per_block + wave_in_block;
  const int comp = (int)(wave_global % kp.n_comp);
  const int64_t tile = wave_global / kp.n_comp;  // this wave's index among the waves of its component
  // PERSISTENT QUADS: frames need very different iteration counts (DexPilot: mean 5, maximum 40+), so a quad that
  // finishes its frame pulls the next one instead of idling until the slowest of the wave's 16 frames is done.  Wave w
  // starts with the static tile [16 w, 16 w + 16); frames from kp.q0 on are numbered by a per-component queue counter
  // that a wave advances by 16 whenever its local pool is empty.
  int64_t item = 0;     // work item (frame, or sequence) this quad is working on
  bool active = false;  // the quad holds a frame (idle quads still execute the passes, on stale data, and store nothing)

  const size_t per_wave = (size_t)64 * (8 * 3 * kp.lds_frames + 4 * 3 * NMAX + 4 * NMAX);
  double* Pl = reinterpret_cast<double*>(lds_raw + (size_t)wave_in_block * per_wave) + lane;
  float* OGl = reinterpret_cast<float*>(lds_raw + (size_t)wave_in_block * per_wave + (size_t)64 * 8 * 3 * kp.lds_frames) + lane;
  // the accepted point (read back only when a trial is rejected): in LDS, the registers it would take are the ones that
  // otherwise spill to scratch
  float* XOl = OGl + (size_t)64 * 3 * NMAX;

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

  uint32_t optmask = 0;
  uint32_t revmask = 0;  // revolute joints: a wave-uniform bit mask, so the loops below test a bit instead of loading jtype
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k < nj && tb.src_kind[k] == DEXR_SRC_OPT) optmask |= 1u << k;
    if (k < nj && tb.jtype[k] == DEXR_JOINT_REVOLUTE) revmask |= 1u << k;
  }

  // ---- per-lane state: replicated vectors + this lane's Hessian rows ------------------------------------------
  float x[NMAX], g[NMAX], d[NMAX];
  float ax[NMAX][3];  // joint origins live in LDS (OGl): only read once per term and chain joint
  float Hq[NHQ];

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
  // ---- load a frame (all four lanes of the quad load the same values) --------------------------------------------
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
#pragma clang loop unroll(disable) vectorize(disable)
  for (int f = 0; f < tb.n_base_frame; ++f) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Pl[(f * 3 + i) * 64] = (double)tb.frame_off[f][i];
  }

  // ---- float64 forward kinematics (replicated) ------------------------------------------------------------------
  auto fk = [&]() {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
    double sR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sp[3] = {0, 0, 0};  // one fork slot (the palm)
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if (k < nj) {
        const int rs = tb.restore[k];
        if (rs == -2) {
          R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
          pp[0] = 0; pp[1] = 0; pp[2] = 0;
        } else if (rs >= 0) {
#pragma unroll
          for (int i = 0; i < 9; ++i) R[i] = sR[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) pp[i] = sp[i];
        }
        const float* Xk = tb.X[k];
#pragma unroll
        for (int i = 0; i < 3; ++i) pp[i] += R[3 * i] * (double)Xk[9] + R[3 * i + 1] * (double)Xk[10] + R[3 * i + 2] * (double)Xk[11];
        double Rn[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = R[3 * i] * (double)Xk[j] + R[3 * i + 1] * (double)Xk[3 + j] + R[3 * i + 2] * (double)Xk[6 + j];
        const double q = (double)x[k];
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
          for (int i = 0; i < 3; ++i) pp[i] += q * Rn[3 * i + 2];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          ax[k][i] = (float)R[3 * i + 2];
          OGl[(k * 3 + i) * 64] = (float)pp[i];
        }
        if (tb.save[k] >= 0) {
#pragma unroll
          for (int i = 0; i < 9; ++i) sR[i] = R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) sp[i] = pp[i];
        }
        const int fb = tb.fbeg[k], fe = tb.fend[k];
#pragma clang loop unroll(disable) vectorize(disable)
        for (int f = fb; f < fe; ++f) {
          const double o0 = tb.frame_off[f][0], o1 = tb.frame_off[f][1], o2 = tb.frame_off[f][2];
#pragma unroll
          for (int i = 0; i < 3; ++i) Pl[(f * 3 + i) * 64] = pp[i] + R[3 * i] * o0 + R[3 * i + 1] * o1 + R[3 * i + 2] * o2;
        }
      }
    }
  };

  const bool per_coord = kp.kind == DEXR_KIND_POSITION;
  const double beta = (double)kp.huber_delta, ibeta = 1.0 / beta;
  const bool newton = kp.newton != 0;

  // ---- fused value / gradient (replicated) / Hessian rows (distributed) at the FK state -----------------------------
  auto assemble = [&]() -> double {
    double Fv = 0;
#pragma unroll
    for (int i = 0; i < NHQ; ++i) Hq[i] = 0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) g[k] = 0;
#pragma clang loop unroll(disable) vectorize(disable)
    for (int t = 0; t < nt; ++t) {
      const int ft = tb.term_task[t], fo = tb.term_origin[t];
      double ptd[3], pod[3] = {0, 0, 0}, rd[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) ptd[i] = Pl[(ft * 3 + i) * 64];
      if (fo >= 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) pod[i] = Pl[(fo * 3 + i) * 64];
      }
      float r[3], pt[3], po[3], fvec[3], hw[3], kap = 0;
      float tv[3], wt;
      term_target(tb.term_ref[t], tv, wt);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        rd[i] = ptd[i] - pod[i] - (double)tv[i];
        r[i] = (float)rd[i];
        pt[i] = (float)ptd[i];
        po[i] = (float)pod[i];
      }
      const double w = (double)kp.inv_norm * (double)wt;
      if (per_coord) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double ee = rd[i], ae = fabs(ee);
          const bool quad = ae < beta;
          Fv += w * (quad ? 0.5 * ee * ee * ibeta : ae - 0.5 * beta);
          fvec[i] = (float)(w * (quad ? ee * ibeta : (ee > 0 ? 1.0 : -1.0)));
          hw[i] = (float)(w * (quad ? ibeta : (kp.newton != 0 ? 0.0 : 1.0 / ae)));  // exact curvature in Newton mode
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
        kap = quad ? 0.f : (float)(psi * id * id);
      }
      const uint32_t mt = tb.frame_anc[ft];
      const uint32_t mo = (fo >= 0) ? tb.frame_anc[fo] : 0u;
      const uint32_t mu = (mt | mo) & optmask;
      float col[NMAX][3];
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if ((mu >> k) & 1u) {
          const bool in_t = (mt >> k) & 1u, in_o = (mo >> k) & 1u;
          if ((revmask >> k) & 1u) {
            float v[3] = {0, 0, 0};
            const float o0 = OGl[(k * 3 + 0) * 64], o1 = OGl[(k * 3 + 1) * 64], o2 = OGl[(k * 3 + 2) * 64];
            if (in_t) { v[0] += pt[0] - o0; v[1] += pt[1] - o1; v[2] += pt[2] - o2; }
            if (in_o) { v[0] -= po[0] - o0; v[1] -= po[1] - o1; v[2] -= po[2] - o2; }
            col[k][0] = ax[k][1] * v[2] - ax[k][2] * v[1];
            col[k][1] = ax[k][2] * v[0] - ax[k][0] * v[2];
            col[k][2] = ax[k][0] * v[1] - ax[k][1] * v[0];
          } else {
            const float sg = (in_t ? 1.f : 0.f) - (in_o ? 1.f : 0.f);
#pragma unroll
            for (int i = 0; i < 3; ++i) col[k][i] = sg * ax[k][i];
          }
          g[k] += col[k][0] * fvec[0] + col[k][1] * fvec[1] + col[k][2] * fvec[2];
        } else {
          col[k][0] = 0; col[k][1] = 0; col[k][2] = 0;
        }
      }
      // this lane's rows r = 4i + p
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if ((mu >> (4 * i)) & 0xFu) {  // some row of this block is on the term's chain (wave-uniform test)
          const float c0 = sel4(p, col[4 * i][0], col[4 * i + 1][0], col[4 * i + 2][0], col[4 * i + 3][0]);
          const float c1 = sel4(p, col[4 * i][1], col[4 * i + 1][1], col[4 * i + 2][1], col[4 * i + 3][1]);
          const float c2 = sel4(p, col[4 * i][2], col[4 * i + 1][2], col[4 * i + 2][2], col[4 * i + 3][2]);
          const uint32_t rt = (mt >> (4 * i + p)) & 1u, ro = (mo >> (4 * i + p)) & 1u;  // lane-varying
          const float ku = kap * (c0 * r[0] + c1 * r[1] + c2 * r[2]);
          const float cw0 = hw[0] * c0 - ku * r[0], cw1 = hw[1] * c1 - ku * r[1], cw2 = hw[2] * c2 - ku * r[2];
          const float cf0 = c1 * fvec[2] - c2 * fvec[1];
          const float cf1 = c2 * fvec[0] - c0 * fvec[2];
          const float cf2 = c0 * fvec[1] - c1 * fvec[0];
#pragma unroll
          for (int c = 0; c < 4 * i + 4; ++c) {
            if ((mu >> c) & 1u) {
              float h = cw0 * col[c][0] + cw1 * col[c][1] + cw2 * col[c][2];
              if (newton && ((revmask >> c) & 1u)) {
                const uint32_t same = (rt & ((mt >> c) & 1u)) | (ro & ((mo >> c) & 1u));
                const float nt2 = ax[c][0] * cf0 + ax[c][1] * cf1 + ax[c][2] * cf2;
                h += same ? nt2 : 0.f;
              }
              Hq[2 * i * (i + 1) + c] += h;  // entries with c > r are never read
            }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if ((optmask >> k) & 1u) {
        const double dx = (double)x[k] - (double)xl(k);
        Fv += (double)delta * dx * dx;
      }
    return Fv;
  };

  // ---- distributed Cholesky + solve (H + mask/damping) d = -g.  freemask is identical in the four lanes. ----------
  auto factor_and_solve = [&](uint32_t freemask, float lam) -> bool {
    bool ok = true;
    // reduced, damped system
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = 4 * i + p;
      const bool fr = (freemask >> r) & 1u;
#pragma unroll
      for (int c = 0; c < 4 * i + 4; ++c) {
        const bool fc = (freemask >> c) & 1u;
        float v = (fr && fc) ? Hq[2 * i * (i + 1) + c] : 0.f;
        if (c >= 4 * i) v = (c == r) ? (fr ? v + 2.f * delta + lam : 1.f) : v;
        Hq[2 * i * (i + 1) + c] = v;
      }
    }
    float ivs[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
      constexpr int dummy = 0;
      (void)dummy;
      const int jo = j >> 2;
      // pivot lives on lane (j & 3), local row jo, column j
      float dj;
      {
        const float dl = Hq[2 * jo * (jo + 1) + j];
        dj = (j & 3) == 0 ? quad_bcast<0>(dl) : (j & 3) == 1 ? quad_bcast<1>(dl) : (j & 3) == 2 ? quad_bcast<2>(dl) : quad_bcast<3>(dl);
      }
      if (!(dj > 1e-30f)) { ok = false; dj = 1.f; }
      const float iv = __frsqrt_rn(dj);
      ivs[j] = iv;
      // scale this lane's part of column j (rows r > j) and remember it
      float lcol[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        lcol[i] = 0.f;
        if (4 * i + 3 > j) {
          const int r = 4 * i + p;
          const float v = Hq[2 * i * (i + 1) + j] * iv;
          const bool below = r > j;
          lcol[i] = below ? v : 0.f;
          Hq[2 * i * (i + 1) + j] = below ? v : Hq[2 * i * (i + 1) + j];
        }
      }
      // trailing update of this lane's rows: H[r][c] -= L[r][j] L[c][j],  j < c <= r
#pragma unroll
      for (int c = j + 1; c < NMAX; ++c) {
        const int co = c >> 2;
        const float lc = (c & 3) == 0 ? quad_bcast<0>(lcol[co]) : (c & 3) == 1 ? quad_bcast<1>(lcol[co])
                       : (c & 3) == 2 ? quad_bcast<2>(lcol[co]) : quad_bcast<3>(lcol[co]);
#pragma unroll
        for (int i = co; i < NR; ++i) Hq[2 * i * (i + 1) + c] -= lcol[i] * lc;  // rows with r < c hold junk there
      }
    }
    // forward: L y = -g (y replicated); row j is complete on lane (j & 3)
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
      const int jo = j >> 2;
      float s = ((freemask >> j) & 1u) ? -g[j] : 0.f;
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Hq[2 * jo * (jo + 1) + k] * d[k];
      s *= ivs[j];
      d[j] = (j & 3) == 0 ? quad_bcast<0>(s) : (j & 3) == 1 ? quad_bcast<1>(s) : (j & 3) == 2 ? quad_bcast<2>(s) : quad_bcast<3>(s);
    }
    // backward: L^T d = y; column j is spread over the quad -> partial sums + quad reduction
#pragma unroll
    for (int j = NMAX - 1; j >= 0; --j) {
      float part = 0.f;
#pragma unroll
      for (int i = (j + 1) >> 2; i < NR; ++i) {
        if (4 * i + 3 > j) {
          const int r = 4 * i + p;
          const float xr = sel4(p, d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
          part += (r > j) ? Hq[2 * i * (i + 1) + j] * xr : 0.f;
        }
      }
      const float tot = quad_sum(part);
      d[j] = (d[j] - tot) * ivs[j];
    }
    return ok;
  };

  // ---- projected Levenberg-Marquardt / Newton (single call site per stage, like dexr_big.hpp) ---------------------
  float lam = kp.lam0, nu = 2.f, sprev = 1e30f, keff = 0.f;
  bool done = true, pending = false;  // done: no solve in progress in this quad (idle, or finished and about to retire)
  int status = ST_MAXITER, my_iters = 0, blind = 0, my_pass = 0;
  double F = 0;
  float smax = 0, pred = 0;
  bool ok = true;
  const int max_pass = 2 * kp.max_iter + 2;
  // wave-uniform pool of unassigned frames
  unsigned pool_next = (unsigned)((tile * 16 < (int64_t)kp.q0 && tile * 16 < nB) ? tile * 16 : 0);
  unsigned pool_end = (unsigned)((tile * 16 < (int64_t)kp.q0 && tile * 16 < nB)
                                     ? ((tile * 16 + 16 < nB) ? tile * 16 + 16 : nB) : 0);
  bool dry = false;  // the queue is exhausted
  unsigned* queue = kp.queue + comp;
  for (;;) {
    // (0) hand frames to idle quads
    const unsigned long long want = __ballot(!active);
    if (want != 0ull) {
      if (pool_next >= pool_end && !dry) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(queue, 16u);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + kp.q0;
        if ((int64_t)base >= nB) {
          dry = true;
        } else {
          pool_next = base;
          pool_end = (unsigned)(((int64_t)base + 16 < nB) ? base + 16 : nB);
        }
      }
      if (pool_next < pool_end) {
        // rank of this quad among the wanting quads = wanting lanes below it / 4 (the four lanes of a quad agree)
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
        const unsigned cand = pool_next + ((below - (unsigned)p) >> 2);
        const bool got = !active && cand < pool_end;
        pool_next += (unsigned)__popcll(__ballot(got)) >> 2;
        if (got) {
          load_frame((int64_t)cand, 0);
          active = true;
          done = false;
          pending = false;
          lam = kp.lam0;
          nu = 2.f;
          sprev = 1e30f;
          keff = 0.f;
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
          // damping jump: at least lam_jump x the curvature of the damped model along the step that just failed
          // (Rayleigh quotient -g.d / d.d; g and d are replicated in the quad, so the four lanes agree bit for bit)
          if (kp.lam_jump > 0) lam = fmaxf(lam, kp.lam_jump * keff);
          nu *= 2.f;
#pragma unroll
          for (int k = 0; k < NMAX; ++k) x[k] = XOl[k * 64];
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
          rebuild = true;
        }
        if (!done && my_iters >= kp.max_iter) done = true;
      }
    }
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
    if (stepping) {
      pred = alpha * (1.f - 0.5f * alpha) * gd + 0.5f * alpha * alpha * lam * dd;
      keff = gd / fmaxf(dd, 1e-30f);
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      if (stepping) {
        XOl[k * 64] = x[k];
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
        for (int k = 0; k < NMAX; ++k) x[k] = XOl[k * 64];
        pending = false;
      }
      bool bad = false;
#pragma unroll
      for (int k = 0; k < NMAX; ++k)
        if ((optmask >> k) & 1u) bad = bad || !(x[k] == x[k]);
      if (bad) status = ST_FALLBACK;
      if (p == 0) {
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
      }
      if (seq && t_seq + 1 < kp.T) {
        // next frame of this quad's sequence: its start point / regularisation target is the row lane 0 of the quad
        // has just written.  The other three lanes read it back with agent-scope loads: program order within a wave
        // is not a memory-model guarantee across lanes, so the stores are released first (as dexr_wide.hpp does)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        load_frame(item, t_seq + 1);
        done = false;
        pending = false;
        lam = kp.lam0;
        nu = 2.f;
        sprev = 1e30f;
        keff = 0.f;
        status = ST_MAXITER;
        my_iters = 0;
        blind = 0;
        my_pass = 0;
        F = 0;
        smax = 0;
        pred = 0;
        ok = true;
      } else {
        if (p == 0 && dexpilot && kp.state && comp == 0) kp.state[lrow] = nst;
        active = false;
      }
    }
  }
}

}  // namespace dexr
