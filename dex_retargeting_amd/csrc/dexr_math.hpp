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

}  // namespace dexr
