// quant.hip -- integer-valued transforms named in the parity clause (K16):
//   MuLaw.encode / decode   reference transforms/raw.py:265-316 (torchaudio MuLawEncoding/Decoding)
//   OneHot.forward / invert reference transforms/misc.py:156-213 (F.one_hot / argmax)
// Integer outputs must match bit for bit, so the mu-law chain keeps torchaudio's
// fp32 operation order (compiled with -ffp-contract=off) and takes log1p in fp64
// rounded once to fp32 (a correctly rounded fp32 log1p).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"

namespace at_hip {

__device__ __forceinline__ float log1p_cr(float v) { return (float)log1p((double)v); }

__global__ void mulaw_encode_kernel(const float* __restrict__ x, long long n, float mu, long long* __restrict__ out) {
  const float l1p_mu = log1p_cr(mu);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float sgn = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
    float y = sgn * log1p_cr(mu * fabsf(v));
    y = y / l1p_mu;
    const float q = (y + 1.0f) / 2.0f * mu + 0.5f;
    out[i] = (long long)q;  // truncation toward zero, like Tensor.to(int64)
  }
}

__global__ void mulaw_decode_kernel(const long long* __restrict__ codes_i, const float* __restrict__ codes_f,
                                    long long n, float mu, float* __restrict__ out) {
  const float l1p_mu = log1p_cr(mu);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float c = codes_i ? (float)codes_i[i] : codes_f[i];
    const float v = (c / mu) * 2.0f - 1.0f;
    const float sgn = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
    out[i] = sgn * ((float)exp((double)(fabsf(v) * l1p_mu)) - 1.0f) / mu;
  }
}

// out[i, c] = (x[i] == c), int64; `channel_major`: out[(i / inner) , c, i % inner]  (one_hot(...).transpose(-1,-2))
__global__ void onehot_kernel(const long long* __restrict__ x, long long n, int classes, long long inner,
                              long long* __restrict__ out) {
  const long long total = n * classes;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (long long)gridDim.x * blockDim.x) {
    if (inner == 0) {
      const long long i = j / classes;
      const int c = (int)(j - i * classes);
      out[j] = (x[i] == c) ? 1 : 0;
    } else {
      // j indexes (outer, c, t) with t < inner
      const long long t = j % inner;
      const long long oc = j / inner;
      const int c = (int)(oc % classes);
      const long long o = oc / classes;
      out[j] = (x[o * inner + t] == c) ? 1 : 0;
    }
  }
}

// first index of the maximum over the last dim (Tensor.argmax(-1))
__global__ void argmax_last_kernel(const long long* __restrict__ xi, const float* __restrict__ xf, long long rows,
                                   int cols, long long* __restrict__ out) {
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
    int best = 0;
    if (xi) {
      long long bv = xi[r * cols];
      for (int c = 1; c < cols; ++c) {
        const long long v = xi[r * cols + c];
        if (v > bv) { bv = v; best = c; }
      }
    } else {
      float bv = xf[r * cols];
      for (int c = 1; c < cols; ++c) {
        const float v = xf[r * cols + c];
        if (v > bv) { bv = v; best = c; }
      }
    }
    out[r] = best;
  }
}

static inline unsigned qgrid(long long n) {
  long long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_mulaw_encode(const float* x, int64_t n, int channels, int64_t* codes, void* stream) {
  if (n < 0 || channels < 2) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!x || !codes) return AT_EINVAL;
  hipLaunchKernelGGL(mulaw_encode_kernel, dim3(qgrid(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n,
                     (float)(channels - 1.0), (long long*)codes);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_mulaw_decode(const int64_t* codes_i64, const float* codes_f32, int64_t n, int channels, float* x, void* stream) {
  if (n < 0 || channels < 2) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if ((!codes_i64 && !codes_f32) || !x) return AT_EINVAL;
  hipLaunchKernelGGL(mulaw_decode_kernel, dim3(qgrid(n)), dim3(256), 0, (hipStream_t)stream, (const long long*)codes_i64,
                     codes_f32, (long long)n, (float)(channels - 1.0), x);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_onehot(const int64_t* x, int64_t n, int classes, int64_t channel_major_inner, int64_t* out, void* stream) {
  if (n < 0 || classes <= 0 || channel_major_inner < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!x || !out) return AT_EINVAL;
  hipLaunchKernelGGL(onehot_kernel, dim3(qgrid(n * classes)), dim3(256), 0, (hipStream_t)stream, (const long long*)x,
                     (long long)n, classes, (long long)channel_major_inner, (long long*)out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_argmax_last(const int64_t* x_i64, const float* x_f32, int64_t rows, int cols, int64_t* out, void* stream) {
  if (rows < 0 || cols <= 0) return AT_EINVAL;
  if (rows == 0) return AT_OK;
  if ((!x_i64 && !x_f32) || !out) return AT_EINVAL;
  hipLaunchKernelGGL(argmax_last_kernel, dim3(qgrid(rows)), dim3(256), 0, (hipStream_t)stream, (const long long*)x_i64,
                     x_f32, (long long)rows, cols, (long long*)out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
