// resample.hip -- band-limited sample-rate conversion for the audio front end (`import_data`).
//
// Replaces torchaudio.transforms.Resample as the reference calls it (utils/misc.py:31-33: default arguments,
// i.e. Hann-windowed sinc interpolation, lowpass_filter_width 6, rolloff 0.99).  torchaudio is not part of the
// reference tree (requirements.txt:3, unpinned), so its published algorithm is restated: with the rates reduced
// by their gcd to (orig, new), every block of `orig` input samples yields `new` output samples,
//     y[i * new + j] = sum_k h[j][k] * xpad[i * orig + k],    k = 0 .. 2 width + orig - 1,
// xpad = x zero-padded by `width` in front and `width + orig` behind, h = the polyphase filter bank the host
// builds in float64 (utils/audio_io.py), rounded to float32.  One thread per output sample; the filter bank and
// the input window of a block of outputs sit in L1/L2 (h is at most a few hundred KB).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"

namespace at_hip {

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, long long rows, long long L,
                                                             int orig, int nw, int width, const float* __restrict__ h,
                                                             long long out_len, float* __restrict__ out) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = blockIdx.y;
  if (o >= out_len) return;
  const long long i = o / nw;
  const int j = (int)(o - i * nw);
  const int taps = 2 * width + orig;
  const float* hj = h + (long long)j * taps;
  const float* xr = x + row * L;
  const long long first = i * orig - width;          // input index of tap 0
  float acc = 0.f;
  for (int k = 0; k < taps; ++k) {
    const long long n = first + k;
    const float v = (n >= 0 && n < L) ? xr[n] : 0.0f;
    acc = fmaf(hj[k], v, acc);
  }
  out[row * out_len + o] = acc;
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_resample_sinc(const float* x, int64_t rows, int64_t L, int orig, int new_, int width, const float* filters,
                     int64_t out_len, float* out, void* stream) {
  if (rows < 0 || L < 0 || orig <= 0 || new_ <= 0 || width < 0 || out_len < 0) return AT_EINVAL;
  if (rows == 0 || out_len == 0) return AT_OK;
  if (!x || !filters || !out) return AT_EINVAL;
  if (rows > 65535) return AT_EUNSUPPORTED;
  hipLaunchKernelGGL(resample_sinc_kernel, dim3((unsigned)((out_len + 255) / 256), (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, x, (long long)rows, (long long)L, orig, new_, width, filters,
                     (long long)out_len, out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
