// phase_repr.hip -- the phase side of the spectral representations: unwrap, instantaneous frequency
// (finite differences of the unwrapped phase) and its inverse (integration along time).
//
// Replaces utils/misc.py:12-26 (unwrap), :65-81 (fdiff_*), :83-104 (fint_*), and the per-row scalings
// of IF.get_if / IF.invert (transforms/spectral_repr.py:318-335, 360-373), fused with the angle of the
// complex spectrum in front and the Normalize affine behind so that the spectrum is read once.
//
// All of these are scans along the frame axis of a (B, T, F) tensor.  One thread owns one (clip, bin) column
// and walks it in frame order -- the reference's arithmetic is sequential in t (torch.cumsum on CPU
// accumulates in double and rounds every prefix to float; fint_central is a Python loop), so the walk
// reproduces it operation by operation (built with -ffp-contract=off).  Adjacent threads own adjacent bins:
// every step of a wavefront reads/writes 64 consecutive elements of one frame row.  Loads do not depend on
// the recurrence and are unrolled ahead of it.  HBM-bound: 8 (complex) or 4 (phase) bytes in, 4 out per bin.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"

namespace at_hip {

constexpr float kPi = 3.14159265358979323846f;        // float(torch.pi)
constexpr float kTwoPi = 6.28318530717958647692f;     // float(2 * torch.pi)

enum { SCAN_UNWRAP = 0, SCAN_IF_FORWARD = 1, SCAN_IF_BACKWARD = 2, SCAN_IF_CENTRAL = 3, SCAN_ANGLE = 4 };

struct ScanParams {
  const float2* X;       // complex spectrum, or
  const float* phase;    // wrapped phase (exactly one of the two)
  float* out;
  long long B, T, F;
  const float* window;   // optional per-frame weight (IF weighted), T floats
  const float* offset;   // optional Normalize affine (device scalars)
  const float* scale;
  int bare;              // 1: plain fdiff_* of a real signal (no unwrap, no per-row division): utils/misc.py:65-81
};

__device__ __forceinline__ float wrapped_phase(const ScanParams& p, long long idx) {
  if (p.X) {
    const float2 z = p.X[idx];
    return atan2f(z.y, z.x);
  }
  return p.phase[idx];
}

// the correction torch's unwrap adds for one frame-to-frame jump (utils/misc.py:19-24)
__device__ __forceinline__ float unwrap_correction(float jump) {
  float r = fmodf(jump + kPi, kTwoPi);               // torch.remainder: result takes the divisor's sign
  if (r != 0.0f && r < 0.0f) r += kTwoPi;
  float folded = r - kPi;
  if (folded == -kPi && jump > 0.0f) folded = kPi;
  const float corr = folded - jump;
  return (fabsf(jump) < kPi) ? 0.0f : corr;
}

template <int MODE>
__global__ __launch_bounds__(256) void phase_scan_kernel(ScanParams p) {
  const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= p.B * p.F) return;
  const long long b = col / p.F, f = col - b * p.F;
  const long long base = b * p.T * p.F + f;
  const long long T = p.T, F = p.F;
  float off = 0.f, sc = 1.f;
  const bool norm = p.offset != nullptr;
  if (norm) {
    off = *p.offset;
    sc = *p.scale;
  }
  auto emit = [&](long long t, float v) {
    if (p.window) v = p.window[t] * v;
    if (norm) v = (v - off) / sc;
    p.out[base + t * F] = v;
  };
  if (MODE == SCAN_ANGLE) {
#pragma unroll 4
    for (long long t = 0; t < T; ++t) emit(t, wrapped_phase(p, base + t * F));
    return;
  }
  float raw_prev = wrapped_phase(p, base);
  double acc = 0.0;                 // torch.cumsum's accumulator on CPU
  float u_prev = raw_prev;          // unwrapped phase of frame t-1
  float u_prev2 = 0.f;              // ... of frame t-2
  if (MODE == SCAN_UNWRAP || MODE == SCAN_IF_CENTRAL) emit(0, raw_prev);
  const bool div = !p.bare;
  if (MODE == SCAN_IF_FORWARD) emit(0, (div && T > 1) ? raw_prev / kPi : raw_prev);   // rows [0, T-2] are divided by pi
#pragma unroll 4
  for (long long t = 1; t < T; ++t) {
    const float raw = wrapped_phase(p, base + t * F);
    if (div) acc += (double)unwrap_correction(raw - raw_prev);
    const float u = raw + (float)acc;
    if (MODE == SCAN_UNWRAP) {
      emit(t, u);
    } else if (MODE == SCAN_IF_FORWARD) {
      const float d = (u - u_prev) / 2.0f;
      emit(t, (div && t < T - 1) ? d / kPi : d);
    } else if (MODE == SCAN_IF_BACKWARD) {
      const float d = (u_prev - u) / 2.0f;            // row t-1; rows >= 1 are divided by -pi
      emit(t - 1, (div && t - 1 >= 1) ? d / (-kPi) : d);
    } else if (MODE == SCAN_IF_CENTRAL) {
      if (t >= 2) {                                                // interior rows
        const float d = (u - u_prev2) / 4.0f;
        emit(t - 1, div ? d / kTwoPi : d);
      }
    }
    raw_prev = raw;
    u_prev2 = u_prev;
    u_prev = u;
  }
  if (MODE == SCAN_IF_BACKWARD) emit(T - 1, (div && T > 1) ? u_prev / (-kPi) : u_prev);   // last row = the phase itself
  if (MODE == SCAN_IF_CENTRAL && T > 1) emit(T - 1, u_prev);
}

// ---- integration (IF.invert) -------------------------------------------------------------------------
struct IntParams {
  const float* y;        // (B, T, F) instantaneous frequency (normalised when offset/scale are given)
  float* out;
  long long B, T, F;
  const float* offset;
  const float* scale;
};

template <int METHOD>   // SCAN_IF_* ; 0 = plain fint (no row scaling) is selected by `rescale = false`
__global__ __launch_bounds__(256) void phase_integrate_kernel(IntParams p, int rescale) {
  const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= p.B * p.F) return;
  const long long b = col / p.F, f = col - b * p.F;
  const long long base = b * p.T * p.F + f;
  const long long T = p.T, F = p.F;
  float off = 0.f, sc = 1.f;
  const bool norm = p.offset != nullptr;
  if (norm) {
    off = *p.offset;
    sc = *p.scale;
  }
  // de-normalised, re-scaled input row t (spectral_repr.py:362-370)
  auto z = [&](long long t) {
    float v = p.y[base + t * F];
    if (norm) v = v * sc + off;
    if (rescale) {
      if (METHOD == SCAN_IF_FORWARD && t < T - 1) v = v * kPi;
      if (METHOD == SCAN_IF_BACKWARD && t >= 1) v = v * (-kPi);
      if (METHOD == SCAN_IF_CENTRAL && t >= 1 && t < T - 1) v = v * kTwoPi;
    }
    return v;
  };
  if (METHOD == SCAN_IF_FORWARD) {
    double acc = 0.0;
#pragma unroll 4
    for (long long t = 0; t < T; ++t) {
      float v = z(t);
      if (t >= 1) v = v * 2.0f;
      acc += (double)v;
      p.out[base + t * F] = (float)acc;
    }
  } else if (METHOD == SCAN_IF_BACKWARD) {
    double acc = 0.0;
#pragma unroll 4
    for (long long t = T - 1; t >= 0; --t) {
      float v = z(t);
      if (t < T - 1) v = v * 2.0f;
      acc += (double)v;
      p.out[base + t * F] = (float)acc;
    }
  } else {
    // fint_central (utils/misc.py:96-104), statement by statement.  Rows the reference never writes stay 0.
    if (T == 1) {
      p.out[base] = z(0);
      return;
    }
    float even = z(0);                       // out[0]
    p.out[base] = even;
    for (long long i = 2; i < T; i += 2) {   // out[i] = out[i-2] + 4 x[i-1]
      even = even + 4.0f * z(i - 1);
      p.out[base + i * F] = even;
    }
    for (long long i = 1; i < T; i += 2) p.out[base + i * F] = 0.0f;
    // out[T-1]: x[T-1] when T is even (the forward chain only touched even rows), else the chain's last value
    float cur = ((T - 1) & 1) ? z(T - 1) : even;
    p.out[base + (T - 1) * F] = cur;
    for (long long i = T - 1; i >= 1; i -= 2) {   // out[i-2] = out[i] - 4 x[i-1]; i = 1 writes row "-1"
      cur = cur - 4.0f * z(i - 1);
      const long long row = (i - 2 >= 0) ? i - 2 : T - 1;
      p.out[base + row * F] = cur;
    }
  }
}

// mag * exp(i phase) (SpectralRepresentation.invert, spectral_repr.py:449-451)
__global__ __launch_bounds__(256) void polar_to_complex_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                                long long n, float2* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s, c;
    sincosf(phase[i], &s, &c);
    const float m = mag[i];
    out[i] = make_float2(m * c, m * s);
  }
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_phase_scan(const float* X_complex, const float* phase, int64_t B, int64_t T, int64_t F, int mode, int bare,
                  const float* frame_window, const float* offset, const float* scale, float* out, void* stream) {
  if (B < 0 || T < 0 || F < 0) return AT_EINVAL;
  if (B * T * F == 0) return AT_OK;
  if ((X_complex == nullptr) == (phase == nullptr) || !out) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (mode < SCAN_UNWRAP || mode > SCAN_ANGLE) return AT_EINVAL;
  if (bare && (mode < SCAN_IF_FORWARD || mode > SCAN_IF_CENTRAL || !phase)) return AT_EINVAL;
  ScanParams p = {(const float2*)X_complex, phase, out, B, T, F, frame_window, offset, scale, bare};
  const long long cols = B * F;
  const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case SCAN_UNWRAP: hipLaunchKernelGGL(phase_scan_kernel<SCAN_UNWRAP>, grid, block, 0, s, p); break;
    case SCAN_IF_FORWARD: hipLaunchKernelGGL(phase_scan_kernel<SCAN_IF_FORWARD>, grid, block, 0, s, p); break;
    case SCAN_IF_BACKWARD: hipLaunchKernelGGL(phase_scan_kernel<SCAN_IF_BACKWARD>, grid, block, 0, s, p); break;
    case SCAN_IF_CENTRAL: hipLaunchKernelGGL(phase_scan_kernel<SCAN_IF_CENTRAL>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(phase_scan_kernel<SCAN_ANGLE>, grid, block, 0, s, p); break;
  }
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_phase_integrate(const float* y, int64_t B, int64_t T, int64_t F, int method, int rescale, const float* offset,
                       const float* scale, float* out, void* stream) {
  if (B < 0 || T < 0 || F < 0) return AT_EINVAL;
  if (B * T * F == 0) return AT_OK;
  if (!y || !out || y == out) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (method < SCAN_IF_FORWARD || method > SCAN_IF_CENTRAL) return AT_EINVAL;
  IntParams p = {y, out, B, T, F, offset, scale};
  const long long cols = B * F;
  const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (method == SCAN_IF_FORWARD) hipLaunchKernelGGL(phase_integrate_kernel<SCAN_IF_FORWARD>, grid, block, 0, s, p, rescale);
  else if (method == SCAN_IF_BACKWARD) hipLaunchKernelGGL(phase_integrate_kernel<SCAN_IF_BACKWARD>, grid, block, 0, s, p, rescale);
  else hipLaunchKernelGGL(phase_integrate_kernel<SCAN_IF_CENTRAL>, grid, block, 0, s, p, rescale);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_polar_to_complex(const float* mag, const float* phase, int64_t n, float* out_complex, void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!mag || !phase || !out_complex) return AT_EINVAL;
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(polar_to_complex_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mag, phase,
                     (long long)n, (float2*)out_complex);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
