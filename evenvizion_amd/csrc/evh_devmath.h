// evh_devmath.h -- device math shared by the detectors: bit-reproducible restatements of the operator's scalar
// helpers (every float / double expression is one IEEE operation at a time: the library is built with
// -ffp-contract=off), so that HIP results can be compared bit for bit with the CPU oracle's.
#pragma once
#include <hip/hip_runtime.h>

// cv::fastAtan2 (degrees): the 7th-order polynomial of core/mathfuncs
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  float ax = fabsf(x), ay = fabsf(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// sin/cos for x in [0, 2*pi]: Cody-Waite reduction by pi/2 + fixed-order polynomial kernels (bit-reproducible)
__device__ __forceinline__ void det_sincos(double x, double* so, double* co) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00;
  const double pio2_lo = 6.07710050650619224932e-11;
  double fn = __builtin_rint(x * two_over_pi);
  int n = (int)fn;
  double r = (x - fn * pio2_hi) - fn * pio2_lo;
  double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  double ks = r + (z * r) * (S1 + z * ps);
  double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double kc = 1.0 - (0.5 * z - z * pc);
  double s, c;
  switch (n & 3) {
    case 0: s = ks; c = kc; break;
    case 1: s = kc; c = -ks; break;
    case 2: s = -ks; c = -kc; break;
    default: s = -kc; c = ks; break;
  }
  *so = s; *co = c;
}

