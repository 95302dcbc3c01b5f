// fastmath.h -- the one transcendental the phase-side kernels are bound by.
#pragma once
#include <hip/hip_runtime.h>

namespace at_hip {

// atan2f in ~24 instructions (ocml's: ~50, and the Polar / IF / unwrap kernels evaluate it 513 times per frame).
//   t = min(|x|,|y|) / max(|x|,|y|) through v_rcp_f32 (1 ulp);  atan t = t + t s P(s), s = t^2, P of degree 7
//   fitted on [0, 1] by tools/fit_atan.py (approximation error 7e-9);  then the octant / quadrant reflections and
//   the sign of y.  Max abs error 2.7e-7 rad over all quadrants and 6 decades of magnitude (same script, emulated
//   fp32 against float64 atan2) -- about one ulp of pi, two orders inside the 1e-5 parity bar.
// Signed zeros behave as in libm (atan2(+-0, -0) = +-pi).  Arguments whose larger magnitude is not a normal
// number below 1e37 (zeros, denormals, huge, inf, nan) take libm's atan2f: v_rcp_f32 flushes denormals.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  if (!(mx >= 1.17549435e-38f && mx <= 1e37f)) return atan2f(y, x);
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float s = t * t;
  float u = 2.622235334e-03f;
  u = fmaf(u, s, -1.513249893e-02f);
  u = fmaf(u, s, 4.112179577e-02f);
  u = fmaf(u, s, -7.366700470e-02f);
  u = fmaf(u, s, 1.057392955e-01f);
  u = fmaf(u, s, -1.418597400e-01f);
  u = fmaf(u, s, 1.999039650e-01f);
  u = fmaf(u, s, -3.333298564e-01f);
  float r = fmaf(u * s, t, t);
  r = (ay > ax) ? 1.57079632679489661923f - r : r;
  r = (__float_as_uint(x) >> 31) ? 3.14159265358979323846f - r : r;
  return copysignf(r, y);
}

// sin / cos of a phase of any size (unwrapped phases reach 1e5 rad): reduced in fp64 to revolutions in [-0.5, 0.5]
// (exact to ~1e-16), then the hardware v_sin_f32 / v_cos_f32 (they take revolutions; abs error ~1e-6, inside the 1e-5
// bar of everything that multiplies a magnitude by them).  ~10 instructions against ~45 for ocml's sincosf.
__device__ __forceinline__ void fast_sincosf(float phase, float& s, float& c) {
  double t = (double)phase * 0.15915494309189533577;  // 1 / (2 pi)
  t -= rint(t);
  const float r = (float)t;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

}  // namespace at_hip
