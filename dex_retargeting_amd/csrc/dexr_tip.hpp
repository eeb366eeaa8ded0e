// dexr_tip.hpp -- the pass of a "tip" component, written out for the chain kernel (dexr_kernel<4, float, SOLVE, CHAIN, EXT, TIP>).
//
// A tip component is what every finger of the per-finger vector models is (teleop Allegro / LEAP, BASELINE.json
// configs[0] and [1], the headline): an unbranched chain of four revolute optimised joints hanging off the base and ONE
// residual term -- the vector from a frame on the fixed base to a frame on the last joint (VectorOptimizer with one
// (origin, task) pair per finger, /root/reference/src/dex_retargeting/optimizer.py:203-306).  The generic pass walks that
// structure through tables: per joint a scalar load of the placement + wait, a rolled loop over the frames attached to it,
// frame positions through LDS, a rolled term loop with three dependent scalar loads, ancestor masks tested bit by bit.  Measured (tools/prof_small_stages.sh, s_memtime): ~6 000 cycles per pass for a lone
// wave at ~550 VALU instructions -- the launch of 65 536 frames is bound by the slowest frame's passes at that latency
// (4 096 frames: 0.044 ms, 65 536: 0.063 ms).  Here the pass is ONE basic block:
//   * the constants of the component pinned in SGPRs for the whole kernel (the compiler cannot turn them back into loads) or
//     broadcast from the wave's LDS, laid out as the register PAIRS the packed instructions take;
//   * forward kinematics in v_pk_fma_f32 form: a rotation is kept as rows (R[i][0], R[i][1]) + R[i][2]; the product with
//     the next placement yields (column 0, column 1) and (axis component, origin component) pairs, 6 packed FMAs per row;
//     the joint rotation Rz(q) is two packed operations per row; the last joint's rotation is never formed (the tip offset
//     is rotated instead);
//   * Jacobian columns, gradient and Hessian for the joint PAIRS (0,1) and (2,3) at once: packed cross products, a Hessian
//     row costs 7 packed FMAs per pair of entries;
//   * sines / cosines of the four joint angles evaluated two at a time, reduced by pi instead of pi/2 (one sign flip
//     instead of a quadrant selection).
// No scalar loads, no branches; LDS only for the target and the broadcast constants.
//
// Round 4: the same pass in FLOAT64 (TipTabT<double> / tip_eval<double>: the reference's own arithmetic type,
// optimizer.py:249-304).  The formulas are the ones above written on a two-component vector type, which is a register
// pair of v_pk_* operands for float and simply two scalars (v_fma_f64 each) for double; constants are not pinned there
// (46+ SGPR pairs do not fit) -- joint 0 / offset / box come from the tables as scalar loads hoisted by the compiler, the
// placements of joints 1..3 are broadcast from LDS as doubles.
#pragma once
#include <hip/hip_runtime.h>

#include "dexr_math.hpp"
#include "dexr_tables.h"

namespace dexr {

typedef float kv2 __attribute__((ext_vector_type(2)));  // a register pair: operands of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
static __device__ __forceinline__ kv2 tip_yx(kv2 v) { return v.yx; }
// (Measured in round 4, tools/valu_rate.hip: with four waves on a SIMD a v_pk_fma_f32 issues in 2.74 cycles against 1.74 for
// a v_fma_f32 -- 21 % cheaper per flop -- and a lone wave issues either every 5.07 cycles, so packing halves the lone-wave
// time; the same pass on plain float pairs ran the headline launch in 46.4 instead of 47.5 us: no difference.)
template <typename R> struct TipVec;
template <> struct TipVec<float> { typedef kv2 v2; };
template <> struct TipVec<double> { typedef double v2 __attribute__((ext_vector_type(2))); };
static __device__ __forceinline__ typename TipVec<double>::v2 tip_yx(typename TipVec<double>::v2 v) { return v.yx; }

// A wave-uniform value the compiler must KEEP (in an SGPR, or a VGPR lane when those run out): the empty asm makes its source
// opaque (a vector register of unknown contents), so the readfirstlane can neither be folded back into the kernel-argument /
// table load the value came from nor be re-loaded inside the pass.
static __device__ __forceinline__ int tip_pin(int v) {
  asm volatile("" : "+v"(v));
  return __builtin_amdgcn_readfirstlane(v);
}
static __device__ __forceinline__ float tip_pin(float v) { return __int_as_float(tip_pin(__float_as_int(v))); }
static __device__ __forceinline__ kv2 tip_splat(float v) { return kv2{v, v}; }
static __device__ __forceinline__ typename TipVec<double>::v2 tip_splat(double v) { return typename TipVec<double>::v2{v, v}; }
// constants of the float32 pass are pinned; the float64 pass takes them as they come (see the header)
template <typename R> static __device__ __forceinline__ R tip_const(float v) {
  if constexpr (sizeof(R) == 4) return tip_pin(v);
  else return (R)v;
}

// Wave-uniform constants of one tip component.  The origin frame's position is folded into the first joint's placement:
// every world position below is relative to it, which leaves the residual, the Jacobian and the Hessian unchanged.
// Joint 0, the tip offset and the box are pinned in SGPRs (23); the placements of joints 1..3 (36 values) would push the
// kernel's scalar state past the 102 SGPRs a wave has (spilled to VGPR lanes: a v_readlane per use, measured 80 per pass), so
// they sit in 144 bytes of the wave's LDS, already paired, and are read where they are used (ds_read_b64 of a wave-uniform
// address: a broadcast, no VALU slot).
template <typename R>
struct TipTabT {
  typedef typename TipVec<R>::v2 v2;
  v2 A0[3];         // joint 0 (its parent is the base, R = I): (X0[i][0], X0[i][1])
  R c0[3];          // X0[i][2]: joint 0's axis
  R p0[3];          // X0's translation - origin frame position
  const v2* xl;     // LDS, joints k = 1..3:  xl[6 (k-1) + j]     = (Xk[j][0], Xk[j][1])   (row j of the placement's rotation)
                    //                        xl[6 (k-1) + 3 + j] = (Xk[j][2], Xk's translation[j])
  R off[3];         // task frame origin in the last joint's frame
  R lo[4], hi[4];
  __device__ __forceinline__ v2 XA(int k1, int j) const { return xl[6 * k1 + j]; }
  __device__ __forceinline__ v2 XB(int k1, int j) const { return xl[6 * k1 + 3 + j]; }

  // `ft`: task frame (on joint 3); `fo`: origin frame on the base, or -1; `lds`: 36 reals of the wave's LDS
  __device__ __forceinline__ void load(const dexr_comp_table& tb, int ft, int fo, R* lds, int lane) {
    if (lane < 36) {
      const int q = lane >> 1, h = lane & 1, k = q / 6 + 1, jj = q % 6;
      const int src = jj < 3 ? 3 * jj + h : (h == 0 ? 3 * (jj - 3) + 2 : 9 + (jj - 3));
      lds[lane] = (R)(&tb.X[0][0])[k * 12 + src];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    xl = reinterpret_cast<const v2*>(lds);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      A0[i] = v2{tip_const<R>(tb.X[0][3 * i]), tip_const<R>(tb.X[0][3 * i + 1])};
      c0[i] = tip_const<R>(tb.X[0][3 * i + 2]);
      p0[i] = tip_const<R>(tb.X[0][9 + i] - (fo >= 0 ? tb.frame_off[fo][i] : 0.f));
      off[i] = tip_const<R>(tb.frame_off[ft][i]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lo[k] = tip_const<R>(tb.lo[k]);
      hi[k] = tip_const<R>(tb.hi[k]);
    }
  }
};
typedef TipTabT<float> TipTab;

// sin / cos of two angles at once.  Reduction by pi (k = rint(a / pi), two-constant Cody-Waite: 3.140625 has 9 significant
// bits, so k * 3.140625 is exact), minimax polynomials of degree 9 / 10 on [-pi/2, pi/2] (fitted in float64; errors
// 2.5e-8 / 3.3e-9 before rounding, 1.2e-7 / 1.1e-7 in float32 arithmetic over |a| < 7: RealTraits<float>::sincos measures
// 2.5e-7 on the same angles), and ONE sign for both: sin(a) = (-1)^k sin(r), cos(a) = (-1)^k cos(r) -- the parity bit of k
// shifted into the sign position and XORed in, instead of the quadrant swap / two conditional negations per angle of the
// pi/2 reduction (14 selects and compares per angle there, 4 integer operations here).
static __device__ __forceinline__ void tip_sincos2(kv2 a, kv2* s, kv2* c) {
  const kv2 t = a * 0.318309886183790672f;
  const kv2 kf = kv2{rintf(t.x), rintf(t.y)};
  kv2 r = a - kf * 3.140625f;
  r = r - kf * 9.6765358467e-04f;
  const kv2 z = r * r;
  const kv2 sp = r + (r * z) * (tip_splat(-1.6666665961e-01f) +
                                z * (tip_splat(8.3332417961e-03f) + z * (tip_splat(-1.9822688286e-04f) + z * 2.6345787831e-06f)));
  const kv2 cp = (tip_splat(1.0f) - z * 0.5f) +
                 (z * z) * (tip_splat(4.1666666076e-02f) +
                            z * (tip_splat(-1.3888812242e-03f) + z * (tip_splat(2.4786036849e-05f) + z * -2.6544504746e-07f)));
  const int sx = (int)((unsigned)(int)kf.x << 31), sy = (int)((unsigned)(int)kf.y << 31);  // parity of k -> sign bit
  *s = kv2{__int_as_float(__float_as_int(sp.x) ^ sx), __int_as_float(__float_as_int(sp.y) ^ sy)};
  *c = kv2{__int_as_float(__float_as_int(cp.x) ^ sx), __int_as_float(__float_as_int(cp.y) ^ sy)};
}

// float64: both angles through the shared Cody-Waite / fdlibm routine (dexr_math.hpp)
static __device__ __forceinline__ void tip_sincos2(typename TipVec<double>::v2 a, typename TipVec<double>::v2* s,
                                                   typename TipVec<double>::v2* c) {
  double s0, c0, s1, c1;
  sincos_f64(a.x, &s0, &c0);
  sincos_f64(a.y, &s1, &c1);
  *s = typename TipVec<double>::v2{s0, s1};
  *c = typename TipVec<double>::v2{c0, c1};
}

// One pass at x: returns the data term (reference "huber_distance", no regulariser), writes its gradient g and the lower
// triangle of its Hessian H (hidx order: 00 | 10 11 | 20 21 22 | 30 31 32 33).  t0..t2: the lane's target (origin folded, see
// TipTab); w: the 'mean' factor; nw: 1 with the second-order kinematic term (Newton), 0 without.
template <typename R>
static __device__ __forceinline__ R tip_eval(const TipTabT<R>& tt, const R (&x)[4], R t0, R t1, R t2,
                                             R beta, R ibeta, R w, R nw,
                                             R (&g)[4], R (&H)[10]) {
  typedef typename TipVec<R>::v2 kv2;  // (shadows the float pair: every formula below is written on this type)
  typedef R RR;
  kv2 sA, cA, sB, cB;  // joints (0,1) and (2,3)
  tip_sincos2(kv2{x[0], x[1]}, &sA, &cA);
  tip_sincos2(kv2{x[2], x[3]}, &sB, &cB);
  const RR sn[4] = {sA.x, sA.y, sB.x, sB.y}, cs[4] = {cA.x, cA.y, cB.x, cB.y};

  // ---- forward kinematics -------------------------------------------------------------------------------------------
  kv2 rp[3];      // (R[i][0], R[i][1]) of the running rotation
  RR r2[3];    // R[i][2]
  RR p[3];     // running origin
  kv2 ao[4][3];   // (axis component i, origin component i) of joint k
  {
    const kv2 sc = kv2{sn[0], -sn[0]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const kv2 a = tt.A0[i];
      rp[i] = a * cs[0] + tip_yx(a) * sc;
      r2[i] = tt.c0[i];
      p[i] = tt.p0[i];
      ao[0][i] = kv2{tt.c0[i], tt.p0[i]};
    }
  }
  kv2 n01[3];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const kv2 bx = tip_splat(rp[i].x), by = tip_splat(rp[i].y), bz = tip_splat(r2[i]);
      n01[i] = bx * tt.XA(k - 1, 0) + by * tt.XA(k - 1, 1) + bz * tt.XA(k - 1, 2);
      kv2 t = bx * tt.XB(k - 1, 0) + by * tt.XB(k - 1, 1) + bz * tt.XB(k - 1, 2);
      t.y += p[i];
      ao[k][i] = t;
    }
    if (k < 3) {
      const kv2 sc = kv2{sn[k], -sn[k]};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        rp[i] = n01[i] * cs[k] + tip_yx(n01[i]) * sc;
        r2[i] = ao[k][i].x;
        p[i] = ao[k][i].y;
      }
    }
  }
  // task frame: origin_3 + (Rn Rz(q3)) off = origin_3 + Rn (Rz off)
  const RR ox = cs[3] * tt.off[0] - sn[3] * tt.off[1], oy = sn[3] * tt.off[0] + cs[3] * tt.off[1];
  RR pt[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) pt[i] = ao[3][i].y + n01[i].x * ox + n01[i].y * oy + ao[3][i].x * tt.off[2];

  // ---- residual (LaneSolver::residuals' vector-norm branch) -------------------------------------------------------------
  const RR r[3] = {pt[0] - t0, pt[1] - t1, pt[2] - t2};
  // SmoothL1 of the vector norm (optimizer.py:272-273); 1-ulp v_sqrt / v_rcp
  const RR d2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  RR d, rs = 0;  // (float64: 1 / d from the hardware estimate + two Newton steps, d = d2 / d; see RealTraits<double>)
  if constexpr (sizeof(R) == 4) {
    d = __builtin_amdgcn_sqrtf(d2);
  } else {
    rs = RealTraits<double, true>::rsqrt(d2);
    d = d2 > 0 ? d2 * rs : (RR)0;
  }
  const bool quad = d < beta;
  const RR F = w * (quad ? (RR)0.5 * d2 * ibeta : d - (RR)0.5 * beta);
  RR id;  // d >= beta > 0 in the linear branch
  if constexpr (sizeof(R) == 4) id = quad ? ibeta : __builtin_amdgcn_rcpf(d);
  else id = quad ? ibeta : rs;
  const RR psi = w * id;
  const RR fvec[3] = {psi * r[0], psi * r[1], psi * r[2]};
  const RR kap = quad ? (RR)0 : psi * id * id;
  const RR fn[3] = {nw * fvec[0], nw * fvec[1], nw * fvec[2]};

  // ---- Jacobian columns, gradient, Hessian: joint pairs A = (0,1), B = (2,3) ----------------------------------------------
  kv2 axA[3], ogA[3], axB[3], ogB[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    axA[i] = kv2{ao[0][i].x, ao[1][i].x};
    ogA[i] = kv2{ao[0][i].y, ao[1][i].y};
    axB[i] = kv2{ao[2][i].x, ao[3][i].x};
    ogB[i] = kv2{ao[2][i].y, ao[3][i].y};
  }
  kv2 colA[3], colB[3], cfA[3], cfB[3], cwA[3], cwB[3];
  kv2 gA, gB, uA, uB;
  auto jac = [&](const kv2 (&axp)[3], const kv2 (&ogp)[3], kv2 (&col)[3], kv2 (&cf)[3], kv2 (&cw)[3], kv2& gp, kv2& up) {
    kv2 v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = tip_splat(pt[i]) - ogp[i];
    col[0] = axp[1] * v[2] - axp[2] * v[1];
    col[1] = axp[2] * v[0] - axp[0] * v[2];
    col[2] = axp[0] * v[1] - axp[1] * v[0];
    gp = col[0] * fvec[0] + col[1] * fvec[1] + col[2] * fvec[2];
    up = col[0] * r[0] + col[1] * r[1] + col[2] * r[2];
    // cf = col x f  (second-order term: d2p/dq_c dq_r . f = a_c . (col_r x f), c an ancestor of r or r itself)
    cf[0] = col[1] * fn[2] - col[2] * fn[1];
    cf[1] = col[2] * fn[0] - col[0] * fn[2];
    cf[2] = col[0] * fn[1] - col[1] * fn[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) cw[i] = col[i] * psi;
  };
  jac(axA, ogA, colA, cfA, cwA, gA, uA);
  jac(axB, ogB, colB, cfB, cwB, gB, uB);
  const kv2 kuA = uA * kap, kuB = uB * kap;
  g[0] = gA.x; g[1] = gA.y; g[2] = gB.x; g[3] = gB.y;

  // entries (r, c) and (r, c + 1) of row r against the joint pair Q = (c, c + 1)
  auto rowpair = [&](RR cw0, RR cw1, RR cw2, RR ku, RR cf0, RR cf1, RR cf2, const kv2 (&colQ)[3],
                     const kv2& uQ, const kv2 (&axQ)[3]) -> kv2 {
    return colQ[0] * cw0 + colQ[1] * cw1 + colQ[2] * cw2 - uQ * ku + axQ[0] * cf0 + axQ[1] * cf1 + axQ[2] * cf2;
  };
  const kv2 q0 = rowpair(cwA[0].x, cwA[1].x, cwA[2].x, kuA.x, cfA[0].x, cfA[1].x, cfA[2].x, colA, uA, axA);
  const kv2 q1 = rowpair(cwA[0].y, cwA[1].y, cwA[2].y, kuA.y, cfA[0].y, cfA[1].y, cfA[2].y, colA, uA, axA);
  const kv2 q2a = rowpair(cwB[0].x, cwB[1].x, cwB[2].x, kuB.x, cfB[0].x, cfB[1].x, cfB[2].x, colA, uA, axA);
  const kv2 q2b = rowpair(cwB[0].x, cwB[1].x, cwB[2].x, kuB.x, cfB[0].x, cfB[1].x, cfB[2].x, colB, uB, axB);
  const kv2 q3a = rowpair(cwB[0].y, cwB[1].y, cwB[2].y, kuB.y, cfB[0].y, cfB[1].y, cfB[2].y, colA, uA, axA);
  const kv2 q3b = rowpair(cwB[0].y, cwB[1].y, cwB[2].y, kuB.y, cfB[0].y, cfB[1].y, cfB[2].y, colB, uB, axB);
  H[0] = q0.x;
  H[1] = q1.x; H[2] = q1.y;
  H[3] = q2a.x; H[4] = q2a.y; H[5] = q2b.x;
  H[6] = q3a.x; H[7] = q3a.y; H[8] = q3b.x; H[9] = q3b.y;
  return F;
}

}  // namespace dexr
