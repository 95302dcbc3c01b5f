// sinebank.hip -- oscillator-bank resynthesis of a magnitude spectrogram ("sinebank" inversion mode).
//
// Replaces STFT.get_sinebank_inversion (transforms/stft.py:180-191, inherited by DGT) and the per-chunk
// RealtimeSTFT / RealtimeDGT.get_sinebank_inversion (stft.py:276-291, dgt.py:356-371).
//
// Offline:  y[b, n] = sum_k env[b, k, n] * sin(2 pi f_k t_n + phi_k),   env = linear interpolation of the
// (max-normalised) magnitudes along time, / (2 pi).  The oscillator matrix S[k, n] does not depend on the clip,
// and within one block of 128 output samples the interpolation touches only a few consecutive frames
// j, j+1, ... j+P-1 (P = 3 for hop >= 128), so
//     y[b, n] = sum_p W_p[n] O_p[b, n] / (max 2 pi),        O_p = x[:, j_block + p, :] @ S
// i.e. P [B x F] . [F x L] contractions against the *same* S on the exact-fp32 matrix cores (mel.hip's
// projection kernel with a per-column-block frame offset), plus a pointwise combine: 2 P B F L flops on MFMA
// instead of B F L sines.
//
// The reference evaluates the oscillator phase in fp32 -- fl(fl(fl(2 pi) f_k) t_n) + phi_k reaches ~5e5 rad at
// the end of a 4 s clip, where one ulp is 1/16 rad -- so S must be built from exactly that rounded argument
// (taken in fp32 below, -ffp-contract=off), then sin() of it is evaluated accurately (fp64).  f_k, t_n and phi_k
// come from the host, computed with the very torch calls the reference makes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"
#include "mel_gemm.h"

namespace at_hip {

__global__ __launch_bounds__(256) void sine_matrix_kernel(const float* __restrict__ c, const float* __restrict__ t,
                                                           const float* __restrict__ phi, int F, long long L,
                                                           float* __restrict__ S) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (n >= L) return;
  float a = c[k] * t[n];     // fp32 product, then fp32 sum: the reference's argument, bit for bit
  a = a + phi[k];
  S[(long long)k * L + n] = (float)sin((double)a);
}

__global__ __launch_bounds__(256) void sine_combine_kernel(const float* __restrict__ O, const float* __restrict__ W,
                                                            const float* __restrict__ max_abs, long long B, long long L,
                                                            int n_pass, float* __restrict__ out) {
  const long long total = B * L;
  const float m = *max_abs;
  const float two_pi = 6.28318530717958647692f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i % L;
    float v = 0.f;
    for (int p = 0; p < n_pass; ++p) {
      const float w = W[p * L + n];
      if (w != 0.0f) v = fmaf(w, O[p * total + i], v);     // passes a sample does not use may hold anything
    }
    out[i] = (v / m) / two_pi;
  }
}

// Realtime: y[s, t, n] = (1/F) sum_k x[s, t, k] sin(c_k tau[t, n] + phi[s, k]); the phase offsets differ per
// stream, so there is no shared oscillator matrix -- one thread per output sample evaluates its F sines
// (argument in fp32 as the reference, reduced to revolutions in fp64, v_sin_f32).
__global__ __launch_bounds__(256) void sinebank_rt_kernel(const float* __restrict__ x, const float* __restrict__ c,
                                                           const float* __restrict__ tau, const float* __restrict__ phi,
                                                           long long S_, int T, int F, int N,
                                                           const float* __restrict__ window, float* __restrict__ out) {
  extern __shared__ float sh[];          // x row, c, phi row: 3 F floats
  float* xs = sh;
  float* cs = sh + F;
  float* ps = sh + 2 * F;
  const long long st = blockIdx.y;       // stream * T + frame
  const long long s = st / T;
  const int t = (int)(st - s * T);
  for (int k = threadIdx.x; k < F; k += blockDim.x) {
    xs[k] = x[st * F + k];
    cs[k] = c[k];
    ps[k] = phi[s * F + k];
  }
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float tn = tau[(long long)t * N + n];
  float acc = 0.f;
  for (int k = 0; k < F; ++k) {
    float a = cs[k] * tn;
    a = a + ps[k];
    double r = (double)a * 0.15915494309189533577;   // revolutions
    r -= rint(r);
    acc = acc + xs[k] * __builtin_amdgcn_sinf((float)r);
  }
  acc = acc / (float)F;
  if (window) acc = acc * window[n];     // invert(mode="sinebank") = frames * inv_window (stft.py:303-304)
  out[st * N + n] = acc;
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

size_t at_sinebank_workspace_bytes(int64_t B, int F, int64_t L, int n_pass) {
  return ((size_t)F * (size_t)L + (size_t)n_pass * (size_t)B * (size_t)L) * sizeof(float) + 256;
}

int at_sinebank_offline(const float* x, int64_t B, int64_t T, int F, const float* c, const float* t, const float* phi,
                        int64_t L, int n_pass, const int64_t* block_frame_offset, const float* W3, const float* max_abs,
                        float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || T <= 0 || F <= 0 || L < 0 || n_pass < 1 || n_pass > 64) return AT_EINVAL;
  if (B == 0 || L == 0) return AT_OK;
  if (!x || !c || !t || !phi || !block_frame_offset || !W3 || !max_abs || !out) return AT_EINVAL;
  if (!workspace || workspace_bytes < at_sinebank_workspace_bytes(B, F, L, n_pass)) return AT_EWORKSPACE;
  if (L > 0x7fffffffLL) return AT_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* S = (float*)workspace;
  float* O = S + (size_t)F * (size_t)L;
  hipLaunchKernelGGL(sine_matrix_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)F), dim3(256), 0, s, c, t, phi, F,
                     (long long)L, S);
  if (hipGetLastError() != hipSuccess) return AT_ELAUNCH;
  for (int p = 0; p < n_pass; ++p) {
    MelParams mp = {};
    mp.A = x; mp.Bm = S; mp.out = O + (size_t)p * (size_t)B * (size_t)L;
    mp.offset = nullptr; mp.scale = nullptr;
    mp.rows = B; mp.lda = T * (int64_t)F; mp.ld_out = L; mp.T = 0;
    mp.K = F; mp.N = (int)L; mp.ldb = (int)L;
    mp.a_kind = A_REAL; mp.contrast = C_NONE; mp.inverse = 0; mp.eps = 0.f;
    mp.dense = 1;
    mp.a_block_offset = (const long long*)block_frame_offset + (size_t)p * (size_t)((L + 127) / 128);
    const int rc = launch_mel_project(mp, s);
    if (rc != AT_OK) return rc;
  }
  long long blocks = (B * L + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(sine_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, O, W3, max_abs, (long long)B,
                     (long long)L, n_pass, out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_sinebank_realtime(const float* x, int64_t S_, int T, int F, int N, const float* c, const float* tau,
                         const float* phi, const float* window_or_null, float* out, void* stream) {
  if (S_ < 0 || T <= 0 || F <= 0 || N <= 0) return AT_EINVAL;
  if (S_ == 0) return AT_OK;
  if (!x || !c || !tau || !phi || !out) return AT_EINVAL;
  if (S_ * T > 65535 || (size_t)F * 3 * sizeof(float) > 48 * 1024) return AT_EUNSUPPORTED;   // grid.y, LDS
  hipLaunchKernelGGL(sinebank_rt_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)(S_ * T)), dim3(256),
                     3 * (size_t)F * sizeof(float), (hipStream_t)stream, x, c, tau, phi, (long long)S_, T, F, N,
                     window_or_null, out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
