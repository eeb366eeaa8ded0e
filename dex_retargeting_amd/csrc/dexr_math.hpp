// dexr_math.hpp -- float64 sin / cos for joint angles, shared by the float64 kinematics of every kernel family.
#pragma once
#include <hip/hip_runtime.h>

namespace dexr {

// Cody-Waite reduction by pi/2 (fdlibm constants) + fdlibm kernel polynomials on [-pi/4, pi/4]; no slow path
// (ocml's sincos carries a Payne-Hanek fallback that costs ~100 VGPRs and private memory inside these kernels).
static __device__ __forceinline__ void sincos_f64(double a, double* s, double* c) {
  const double kf = rint(a * 6.36619772367581382433e-01);
  double r = fma(-kf, 1.57079632673412561417e+00, a);
  r = fma(-kf, 6.07710050650619224932e-11, r);
  r = fma(-kf, 2.02226624879595063154e-21, r);
  const double z = r * r;
  const double sp = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
  const double cp = 1.0 - 0.5 * z + z * z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const int k = (int)kf;
  const double ss = (k & 1) ? cp : sp;
  const double cc = (k & 1) ? sp : cp;
  *s = (k & 2) ? -ss : ss;
  *c = ((k + 1) & 2) ? -cc : cc;
}

// sin / cos by TABLE + short polynomials (round 6): a = k pi/16 + r, |r| <= pi/32; (sin, cos)(k pi/16) from a 32-entry table
// (512 B, read through the vector L1: the index is lane-varying), sin r / cos r - 1 from degree-9 / -10 Taylor sums (truncation
// 2e-19 / 2e-21 at |r| = pi/32), recombined by the angle-sum formulas around the table values so that nothing cancels.
// ~22 float64 operations against ~58 + the quadrant selects of sincos_f64 above (two per lane and pass in the sixteen-lane
// kernel: 13 percent of its issue slots); maximum error against mpmath over |a| < 8: 1.1e-16 (same sequence in numpy).
// Two-constant Cody-Waite: PI16_HI has 33 significant bits, k <= 2^7 for |a| < 25: k * PI16_HI is exact.
static __device__ const double SINCOS_TAB16[32][2] = {
    {0.0, 1.0},
    {0.19509032201612828, 0.9807852804032304},
    {0.3826834323650898, 0.9238795325112867},
    {0.5555702330196022, 0.8314696123025452},
    {0.7071067811865476, 0.7071067811865476},
    {0.8314696123025452, 0.5555702330196022},
    {0.9238795325112867, 0.3826834323650898},
    {0.9807852804032304, 0.19509032201612828},
    {1.0, 0.0},
    {0.9807852804032304, -0.19509032201612828},
    {0.9238795325112867, -0.3826834323650898},
    {0.8314696123025452, -0.5555702330196022},
    {0.7071067811865476, -0.7071067811865476},
    {0.5555702330196022, -0.8314696123025452},
    {0.3826834323650898, -0.9238795325112867},
    {0.19509032201612828, -0.9807852804032304},
    {0.0, -1.0},
    {-0.19509032201612828, -0.9807852804032304},
    {-0.3826834323650898, -0.9238795325112867},
    {-0.5555702330196022, -0.8314696123025452},
    {-0.7071067811865476, -0.7071067811865476},
    {-0.8314696123025452, -0.5555702330196022},
    {-0.9238795325112867, -0.3826834323650898},
    {-0.9807852804032304, -0.19509032201612828},
    {-1.0, 0.0},
    {-0.9807852804032304, 0.19509032201612828},
    {-0.9238795325112867, 0.3826834323650898},
    {-0.8314696123025452, 0.5555702330196022},
    {-0.7071067811865476, 0.7071067811865476},
    {-0.5555702330196022, 0.8314696123025452},
    {-0.3826834323650898, 0.9238795325112867},
    {-0.19509032201612828, 0.9807852804032304}};
// `tab`: the table above, or a copy of it (the sixteen-lane kernel keeps one in each wave's LDS)
static __device__ __forceinline__ void sincos_f64_tab(double a, double* s, double* c, const double* tab = &SINCOS_TAB16[0][0]) {
  const double kf = rint(a * 5.092958178940651);
  double r = fma(-kf, 0.1963495408417657, a);
  r = fma(-kf, 7.59637563313274e-12, r);
  const int k = (int)kf & 31;
  const double S = tab[2 * k], C = tab[2 * k + 1];
  const double z = r * r;
  const double sr = fma(r * z, fma(z, fma(z, fma(z, 2.7557319223985893e-06, -1.984126984126984e-04), 8.333333333333333e-03), -1.6666666666666666e-01), r);
  const double cm = z * fma(z, fma(z, fma(z, fma(z, -2.755731922398589e-07, 2.48015873015873e-05), -1.388888888888889e-03), 4.1666666666666664e-02), -0.5);
  *s = S + fma(S, cm, C * sr);
  *c = C + fma(C, cm, -(S * sr));
}

}  // namespace dexr
