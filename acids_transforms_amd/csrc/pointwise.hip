// pointwise.hip -- small elementwise kernels around the spectral path.
//   angle            x_fft.angle()                 reference stft.py:103, dgt.py:69, dgt.py:336   (K2)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"

namespace at_hip {

__global__ void angle_kernel(const float2* __restrict__ x, long long n, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float2 v = x[i];
    out[i] = atan2f(v.y, v.x);
  }
}

static inline unsigned grid_for(long long n, int block) {
  long long b = (n + block - 1) / block;
  if (b > 256 * 8) b = 256 * 8;  // grid-stride above 8 blocks per CU
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_angle(const float* x_complex, int64_t n, float* out, void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!x_complex || !out) return AT_EINVAL;
  hipLaunchKernelGGL(angle_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)x_complex,
                     (long long)n, out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
