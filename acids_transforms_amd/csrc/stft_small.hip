// stft_small.hip -- n_fft = 256 and 128 on the one-wavefront register FFT core (fft512.h): K = 4 / 8 frames share one
// 512-point complex FFT (stft512.hip is the K = 2 case of the same idea, with its own fused overlap-add).
//
// Replaces, for n_fft = 1024 / K:  torch.stft(...).transpose(-2,-1)  (reference transforms/stft.py:98-104,
// dgt.py:64-70), rfft(x*window) on frames (stft.py:249-253, dgt.py:285-289), irfft(X)*inv_window (stft.py:260-266,
// dgt.py:296-302; the frames of torch.istft, overlap-added by stft_generic.hip's gather).  Until round 2 these sizes
// ran on the workgroup-per-frame LDS kernel of stft_generic.hip (2.8 M / 5.6 M workgroups for 1024 clips).
//
// Frame r = 0 .. K-1 of a group has the packed complex sequence a_r[n] = x[2n] + i x[2n+1], n < M = 512 / K.  With
// y[K n + r] = a_r[n] (lane l loads frame l % K, sample (l / K) + (64 / K) j),
//     Y[k + M q] = sum_r W512^(r k) W_K^(r q) A_r[k],           k < M, q < K
// and k + M q is register m0 + (8 / K) q of lane k mod 64 (m0 = k / 64): a K-point inverse DFT across registers of
// one lane recovers K W512^(r k) A_r[k] for every r -- after it EVERY lane holds its bins of all K spectra, so each
// frame's row is stored by a full wave.  Then the real split of each frame (partner A_r[M - k]: lane 64 - lane), as
// in the other sizes.  The inverse runs the same steps backwards.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"
#include "fastmath.h"
#include "fft512.h"

namespace at_hip {

constexpr int WS = 4;   // waves per block

struct PSm {
  const float* x;
  const float* window;   // N samples
  const float2* tw;      // fft512 twiddle table
  const float2* twk;     // W512^(r k): [(r - 1) * M + k], r = 1 .. K-1, k < M
  float2* X;             // (frames, M + 1)
  const float* mag;
  const float* phase;
  float* phase_out;
  float* y;              // inverse: (frames, N)
  long long L, clip_stride, T, total_frames, groups_per_block;
  int hop, center;
};

__device__ __forceinline__ long long reflect_sm(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// K-point DFT across K registers: forward (INV = false): v[q] <- sum_r v[r] W_K^(r q); inverse: conjugate twiddles
template <int K, bool INV>
__device__ __forceinline__ void dftK(v2f (&v)[K]) {
  if constexpr (K == 8) {
    radix8<INV>(v);
  } else {
    static_assert(K == 4, "K = 4 or 8");
    const v2f a0 = v[0] + v[2], a1 = v[0] - v[2];
    const v2f b0 = v[1] + v[3], b1 = v[1] - v[3];
    v[0] = a0 + b0;
    v[2] = a0 - b0;
    v[1] = rot_add<INV>(a1, b1);      // a1 + W4 b1   (W4 = -i forward, +i inverse)
    v[3] = rot_sub<INV>(a1, b1);      // a1 - W4 b1
  }
}

template <int K>
__device__ __forceinline__ void load_lane_frame(const PSm& p, long long f, int lane, float2 (&q)[8]) {
  constexpr int N = 1024 / K, PER = 64 / K;
  if (f >= p.total_frames) {
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = make_float2(0.f, 0.f);
    return;
  }
  const long long b = f / p.T, t = f - b * p.T;
  const float* clip = p.x + b * p.clip_stride;
  const long long start = t * (long long)p.hop - (p.center ? N / 2 : 0);
  const bool interior = (start >= 0) && (start + N <= p.L);
  const int u = lane / K;
  if (interior && ((((uintptr_t)(clip + start)) & 7) == 0)) {
    const float2* src = reinterpret_cast<const float2*>(clip + start);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = src[u + PER * j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long i0 = start + 2 * (u + PER * j);
      float v[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const long long i = i0 + c;
        if (interior) v[c] = clip[i];
        else if (p.center) v[c] = clip[reflect_sm(i, p.L)];
        else v[c] = (i >= 0 && i < p.L) ? clip[i] : 0.0f;     // zero padding past the end (utils/misc.py:156)
      }
      q[j] = make_float2(v[0], v[1]);
    }
  }
}

// mirror partner of A[k], k = lane + 64 m0 (m0 < Q): A[(M - k) mod M]
template <int Q>
__device__ __forceinline__ void mirror_small(const v2f (&v)[Q], v2f (&p)[Q], int lane) {
  const int src = (64 - lane) & 63;
  v2f q[Q];
#pragma unroll
  for (int m = 0; m < Q; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < Q; ++m) {
    const v2f a = q[Q - 1 - m];
    const v2f b = q[(Q - m) % Q];
    p[m] = (lane == 0) ? b : a;
  }
}

template <int K, bool WRITE_PHASE>
__global__ __launch_bounds__(64 * WS) void stft_small_fwd_kernel(PSm p) {
  constexpr int M = 512 / K, Q = 8 / K, F = M + 1, PER = 64 / K;
  __shared__ float2 lds_all[WS * kFftLdsFloat2PerWave];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  Twiddles tw;
  load_twiddles<false>(tw, p.tw, lane);
  v2f wk[K - 1][Q];                              // W512^(r k), r = 1 .. K-1, k = lane + 64 m0
#pragma unroll
  for (int r = 1; r < K; ++r)
#pragma unroll
    for (int m0 = 0; m0 < Q; ++m0) wk[r - 1][m0] = to_v(p.twk[(r - 1) * M + lane + 64 * m0]);
  float2 win[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) win[j] = reinterpret_cast<const float2*>(p.window)[lane / K + PER * j];
  const long long n_groups = (p.total_frames + K - 1) / K;
  const long long g_begin = (long long)blockIdx.x * p.groups_per_block;
  long long g_end = g_begin + p.groups_per_block;
  if (g_end > n_groups) g_end = n_groups;
  const float inv2k = 0.5f / (float)K;

  long long g = g_begin + wave;
  float2 nxt[8];
  if (g < g_end) load_lane_frame<K>(p, K * g + (lane % K), lane, nxt);
  for (; g < g_end; g += WS) {
    v2f y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (v2f){nxt[j].x * win[j].x, nxt[j].y * win[j].y};
    if (g + WS < g_end) load_lane_frame<K>(p, K * (g + WS) + (lane % K), lane, nxt);
    fft512<false>(y, tw, lds, lane);
    // per m0: inverse K-point DFT over q of Y[k + M q] = K W512^(r k) A_r[k]; H_r = A_r / 2
    v2f h[K][Q];
#pragma unroll
    for (int m0 = 0; m0 < Q; ++m0) {
      v2f c[K];
#pragma unroll
      for (int q = 0; q < K; ++q) c[q] = y[m0 + Q * q];
      dftK<K, true>(c);
      h[0][m0] = c[0] * (v2f){inv2k, inv2k};
#pragma unroll
      for (int r = 1; r < K; ++r) h[r][m0] = cmul_conj_v(c[r], wk[r - 1][m0]) * (v2f){inv2k, inv2k};
    }
    // real split of every frame: X[k] = (H + conj H') - i W_N^k (H - conj H'),  W_N^k = W512^((K/2) k)
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const long long f = K * g + r;
      if (f >= p.total_frames) break;            // wave-uniform
      v2f pm[Q];
      mirror_small<Q>(h[r], pm, lane);
      float2* row = p.X + f * F;
#pragma unroll
      for (int m0 = 0; m0 < Q; ++m0) {
        const v2f e = add_conj(h[r][m0], pm[m0]);
        const v2f d = sub_conj(h[r][m0], pm[m0]);
        const v2f xk = add_mi(e, cmul_v(d, wk[K / 2 - 1][m0]));
        row[lane + 64 * m0] = to_f2(xk);
        if (WRITE_PHASE) p.phase_out[f * F + lane + 64 * m0] = fast_atan2f(xk.y, xk.x);
      }
      if (lane == 0) {
        const float ny = 2.0f * (h[r][0].x - h[r][0].y);     // X[M] = Re A[0] - Im A[0]
        row[M] = make_float2(ny, 0.0f);
        if (WRITE_PHASE) p.phase_out[f * F + M] = fast_atan2f(0.0f, ny);
      }
    }
  }
}

__device__ __forceinline__ void sincos_big_sm(float phase, float& s, float& c) {
  double t = (double)phase * 0.15915494309189533577;  // 1 / (2 pi)
  t -= rint(t);
  const float r = (float)t;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

template <int K, bool POLAR>
__global__ __launch_bounds__(64 * WS) void irfft_small_frames_kernel(PSm p) {
  constexpr int N = 1024 / K, M = 512 / K, Q = 8 / K, F = M + 1, PER = 64 / K;
  __shared__ float2 lds_all[WS * kFftLdsFloat2PerWave];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  Twiddles tw;
  load_twiddles<true>(tw, p.tw, lane);
  v2f wk[K - 1][Q];
#pragma unroll
  for (int r = 1; r < K; ++r)
#pragma unroll
    for (int m0 = 0; m0 < Q; ++m0) wk[r - 1][m0] = to_v(p.twk[(r - 1) * M + lane + 64 * m0]);
  const float scale = 1.0f / 1024.0f;           // 1 / 512 of the FFT, 1 / 2 of the split (E + i O = 2 A), for every K
  float2 win[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float2 w = reinterpret_cast<const float2*>(p.window)[lane / K + PER * j];
    win[j] = make_float2(w.x * scale, w.y * scale);
  }
  const long long n_groups = (p.total_frames + K - 1) / K;
  const long long g_begin = (long long)blockIdx.x * p.groups_per_block;
  long long g_end = g_begin + p.groups_per_block;
  if (g_end > n_groups) g_end = n_groups;
  for (long long g = g_begin + wave; g < g_end; g += WS) {
    v2f a[K][Q];                                 // 2 A_r[k], then G_r = W512^(r k) 2 A_r
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const long long f = K * g + r;
      if (f >= p.total_frames) {                 // wave-uniform
#pragma unroll
        for (int m0 = 0; m0 < Q; ++m0) a[r][m0] = (v2f){0.f, 0.f};
        continue;
      }
      v2f v[Q];
      float nyq_re;
      if (POLAR) {
        const float* mrow = p.mag + f * F;
        const float* prow = p.phase + f * F;
#pragma unroll
        for (int m0 = 0; m0 < Q; ++m0) {
          float sn, cs;
          const float gmag = mrow[lane + 64 * m0];
          sincos_big_sm(prow[lane + 64 * m0], sn, cs);
          v[m0] = (v2f){gmag * cs, gmag * sn};
        }
        float sn, cs;
        sincos_big_sm(prow[M], sn, cs);
        nyq_re = mrow[M] * cs;
      } else {
        const float2* row = p.X + f * F;
#pragma unroll
        for (int m0 = 0; m0 < Q; ++m0) v[m0] = to_v(row[lane + 64 * m0]);
        nyq_re = row[M].x;
      }
      if (lane == 0) v[0].y = 0.0f;               // c2r ignores the imaginary parts of DC and Nyquist
      v2f pm[Q];
      mirror_small<Q>(v, pm, lane);
      if (lane == 0) pm[0] = (v2f){nyq_re, 0.0f};
#pragma unroll
      for (int m0 = 0; m0 < Q; ++m0) {
        const v2f e = add_conj(v[m0], pm[m0]);
        const v2f d = cmul_conj_v(sub_conj(v[m0], pm[m0]), wk[K / 2 - 1][m0]);
        const v2f two_a = add_pi(e, d);
        a[r][m0] = (r == 0) ? two_a : cmul_v(two_a, wk[r > 0 ? r - 1 : 0][m0]);
      }
    }
    // Y[k + M q] = sum_r W_K^(r q) G_r[k]: forward K-point DFT across r
    v2f y[8];
#pragma unroll
    for (int m0 = 0; m0 < Q; ++m0) {
      v2f c[K];
#pragma unroll
      for (int r = 0; r < K; ++r) c[r] = a[r][m0];
      dftK<K, false>(c);
#pragma unroll
      for (int q = 0; q < K; ++q) y[m0 + Q * q] = c[q];
    }
    fft512<true>(y, tw, lds, lane);
    const long long f = K * g + (lane % K);
    if (f < p.total_frames) {
      float2* dst = reinterpret_cast<float2*>(p.y + f * N);
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[lane / K + PER * j] = make_float2(y[j].x * win[j].x, y[j].y * win[j].y);
    }
  }
}

static long long groups_per_block_sm(long long ngroups) {
  const long long max_blocks = 256LL * 8;
  long long gpb = (ngroups + max_blocks - 1) / max_blocks;
  gpb = ((gpb + WS - 1) / WS) * WS;
  return gpb < WS ? WS : gpb;
}

int launch_stft_small_fwd(int n_fft, const float* x, long long B, long long L, long long clip_stride, long long T, int hop,
                          int center, const float* window, const float2* tw, const float2* twk, float2* out, float* phase,
                          hipStream_t stream) {
  const long long nframes = B * T;
  if (nframes == 0) return 0;
  const int K = 1024 / n_fft;
  PSm p = {};
  p.x = x; p.window = window; p.tw = tw; p.twk = twk; p.X = out; p.phase_out = phase;
  p.L = L; p.clip_stride = clip_stride; p.T = T; p.total_frames = nframes; p.hop = hop; p.center = center;
  const long long ngroups = (nframes + K - 1) / K;
  p.groups_per_block = groups_per_block_sm(ngroups);
  const unsigned blocks = (unsigned)((ngroups + p.groups_per_block - 1) / p.groups_per_block);
  if (K == 4) {
    if (phase) hipLaunchKernelGGL((stft_small_fwd_kernel<4, true>), dim3(blocks), dim3(64 * WS), 0, stream, p);
    else hipLaunchKernelGGL((stft_small_fwd_kernel<4, false>), dim3(blocks), dim3(64 * WS), 0, stream, p);
  } else if (K == 8) {
    if (phase) hipLaunchKernelGGL((stft_small_fwd_kernel<8, true>), dim3(blocks), dim3(64 * WS), 0, stream, p);
    else hipLaunchKernelGGL((stft_small_fwd_kernel<8, false>), dim3(blocks), dim3(64 * WS), 0, stream, p);
  } else {
    return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft_small_frames(int n_fft, const float2* X, const float* mag, const float* phase, long long nframes,
                              const float* window, const float2* tw, const float2* twk, float* frames, hipStream_t stream) {
  if (nframes == 0) return 0;
  const int K = 1024 / n_fft;
  PSm p = {};
  p.X = const_cast<float2*>(X); p.mag = mag; p.phase = phase; p.window = window; p.tw = tw; p.twk = twk; p.y = frames;
  p.total_frames = nframes;
  const long long ngroups = (nframes + K - 1) / K;
  p.groups_per_block = groups_per_block_sm(ngroups);
  const unsigned blocks = (unsigned)((ngroups + p.groups_per_block - 1) / p.groups_per_block);
  const bool polar = (X == nullptr);
  if (K == 4) {
    if (polar) hipLaunchKernelGGL((irfft_small_frames_kernel<4, true>), dim3(blocks), dim3(64 * WS), 0, stream, p);
    else hipLaunchKernelGGL((irfft_small_frames_kernel<4, false>), dim3(blocks), dim3(64 * WS), 0, stream, p);
  } else if (K == 8) {
    if (polar) hipLaunchKernelGGL((irfft_small_frames_kernel<8, true>), dim3(blocks), dim3(64 * WS), 0, stream, p);
    else hipLaunchKernelGGL((irfft_small_frames_kernel<8, false>), dim3(blocks), dim3(64 * WS), 0, stream, p);
  } else {
    return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
