// stft_mixed.hip -- STFT frames / inverse frames for the FFT sizes that are NOT a power of two (the reference takes
// any n_fft: transforms/stft.py:67-75 hands it to torch.stft; 400, 441, 1000, 1200, 1920, 2000 are everyday values
// at 16 / 44.1 / 48 kHz).  One workgroup per frame at a time, mixed-radix Stockham autosort in LDS: radix 4 / 2 / 3 / 5 / 7
// butterflies in registers, any other prime factor p by a direct p-point DFT per output (O(p) per point: slow for a
// large prime, correct for every size).  Even n_fft: the usual half-size complex transform + split; odd n_fft: a
// full-size complex transform of the real frame.  Correctness path, like stft_generic.hip; the power-of-two sizes
// keep their own kernels.
#include <hip/hip_runtime.h>
#include "fastmath.h"
#include <stdint.h>

namespace at_hip {

constexpr int kMaxMixStages = 16;
struct MixPlan {
  int n_stages;
  int radix[kMaxMixStages];   // product = M
};

__device__ __forceinline__ long long mx_reflect(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}
__device__ __forceinline__ float2 mx_cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 mx_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// exp(sign * 2 pi i idx / M), idx < M: from the LDS table when there is one (M <= 4096)
__device__ __forceinline__ float2 mx_tw(const float2* tw, int idx, int M, float sign) {
  if (tw) return tw[idx];
  float s, c;
  sincospif(sign * 2.0f * (float)idx / (float)M, &s, &c);
  return make_float2(c, s);
}

__device__ void mx_fill_twiddles(float2* tw, int M, float sign) {
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    float s, c;
    sincospif(sign * 2.0f * (float)j / (float)M, &s, &c);
    tw[j] = make_float2(c, s);
  }
}

// one Stockham stage of radix P (compile-time): P inputs in registers, P outputs
template <int P>
__device__ __forceinline__ void mx_stage(const float2* a, float2* b, int M, int Ns, float sign, const float2* tw) {
  const int q_len = M / P;               // butterflies in this stage
  const int tstep = q_len / Ns;          // exp(sign 2 pi i r k / (P Ns)) = tw[r k tstep]
  for (int j = threadIdx.x; j < q_len; j += blockDim.x) {
    const int k = j % Ns;
    float2 v[P];
    v[0] = a[j];
#pragma unroll
    for (int r = 1; r < P; ++r) v[r] = mx_cmul(a[j + r * q_len], mx_tw(tw, r * k * tstep, M, sign));
    const int j0 = (j - k) * P + k;
    if constexpr (P == 2) {
      b[j0] = mx_add(v[0], v[1]);
      b[j0 + Ns] = make_float2(v[0].x - v[1].x, v[0].y - v[1].y);
    } else if constexpr (P == 4) {
      const float2 s02 = mx_add(v[0], v[2]), d02 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
      const float2 s13 = mx_add(v[1], v[3]), d13 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
      const float2 rot = make_float2(-sign * d13.y, sign * d13.x);      // sign * i * d13
      b[j0] = mx_add(s02, s13);
      b[j0 + Ns] = mx_add(d02, rot);
      b[j0 + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
      b[j0 + 3 * Ns] = make_float2(d02.x - rot.x, d02.y - rot.y);
    } else {
      // W_P^(q r) = tw[((q r) mod P) * (M / P)]: P - 1 distinct roots, read once
      float2 w[P];
#pragma unroll
      for (int m = 1; m < P; ++m) w[m] = mx_tw(tw, m * q_len, M, sign);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        float2 acc = v[0];
#pragma unroll
        for (int r = 1; r < P; ++r) {
          const int m = (q * r) % P;
          acc = mx_add(acc, m == 0 ? v[r] : mx_cmul(v[r], w[m]));
        }
        b[j0 + q * Ns] = acc;
      }
    }
  }
}

// any other (prime) radix: every output is a direct p-term sum over LDS, twiddle and root folded into one table entry
__device__ void mx_stage_any(const float2* a, float2* b, int M, int Ns, int p, float sign, const float2* tw) {
  const int q_len = M / p;
  const int tstep = q_len / Ns;
  for (int jq = threadIdx.x; jq < M; jq += blockDim.x) {     // one thread per output (j, q)
    const int j = jq % q_len, q = jq / q_len;
    const int k = j % Ns;
    const long long base = (long long)k * tstep + (long long)q * q_len;   // r * base / M turns
    float2 acc = a[j];
    for (int r = 1; r < p; ++r) acc = mx_add(acc, mx_cmul(a[j + r * q_len], mx_tw(tw, (int)((r * base) % M), M, sign)));
    b[(j - k) * p + k + q * Ns] = acc;
  }
}

// M-point complex FFT of `a` (scratch `b`), factors from the plan.  Returns the buffer holding the result.
__device__ float2* mx_fft(float2* a, float2* b, int M, float sign, const float2* tw, const MixPlan& plan) {
  int Ns = 1;
  for (int s = 0; s < plan.n_stages; ++s) {
    const int p = plan.radix[s];
    switch (p) {
      case 2: mx_stage<2>(a, b, M, Ns, sign, tw); break;
      case 3: mx_stage<3>(a, b, M, Ns, sign, tw); break;
      case 4: mx_stage<4>(a, b, M, Ns, sign, tw); break;
      case 5: mx_stage<5>(a, b, M, Ns, sign, tw); break;
      case 7: mx_stage<7>(a, b, M, Ns, sign, tw); break;
      default: mx_stage_any(a, b, M, Ns, p, sign, tw); break;
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
    Ns *= p;
  }
  return a;
}

struct MixFwdParams {
  const float* x;
  const float* window;
  float2* out;
  float* phase;
  long long B, L, clip_stride, T;
  int n_fft, hop, center, use_tw;
  MixPlan plan;
};

// exp(sign * 2 pi i k / n_fft), k = 0 .. M: the twiddle of the real split (even sizes), once per workgroup
__device__ void mx_fill_split(float2* sp, int M, float sign) {
  for (int k = threadIdx.x; k <= M; k += blockDim.x) {
    float s, c;
    sincospif(sign * (float)k / (float)M, &s, &c);
    sp[k] = make_float2(c, s);
  }
}

// A workgroup walks frames blockIdx.x + k gridDim.x with its tables filled once.
__global__ void rfft_mixed_kernel(MixFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  const int Nf = p.n_fft;
  const bool even = (Nf & 1) == 0;
  const int M = even ? Nf / 2 : Nf;
  float2* a = sm;
  float2* b = sm + M;
  float2* tw = p.use_tw ? sm + 2 * M : nullptr;
  float2* sp = (p.use_tw && even) ? tw + M : nullptr;
  if (tw) mx_fill_twiddles(tw, M, -1.0f);
  if (sp) mx_fill_split(sp, M, -1.0f);
  const int Fb = Nf / 2 + 1;
  for (long long f = blockIdx.x; f < p.B * p.T; f += gridDim.x) {
    const long long bidx = f / p.T, t = f - bidx * p.T;
    const float* clip = p.x + bidx * p.clip_stride;
    const long long start = t * (long long)p.hop - (p.center ? Nf / 2 : 0);
    auto sample = [&](int n) -> float {
      const long long i = start + n;
      const float v = p.center ? clip[mx_reflect(i, p.L)] : (i < p.L ? clip[i] : 0.f);
      return v * p.window[n];
    };
    if (even) {
      for (int n = threadIdx.x; n < M; n += blockDim.x) a[n] = make_float2(sample(2 * n), sample(2 * n + 1));
    } else {
      for (int n = threadIdx.x; n < M; n += blockDim.x) a[n] = make_float2(sample(n), 0.f);
    }
    __syncthreads();
    const float2* Z = mx_fft(a, b, M, -1.0f, tw, p.plan);
    float2* row = p.out + f * Fb;
    float* prow = p.phase ? p.phase + f * Fb : nullptr;
    for (int k = threadIdx.x; k < Fb; k += blockDim.x) {
      float2 X;
      if (even) {
        const float2 zk = Z[k == M ? 0 : k];
        float2 zp = Z[k == 0 ? 0 : M - k];
        zp.y = -zp.y;
        const float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y + zp.y));
        const float2 d = make_float2(0.5f * (zk.x - zp.x), 0.5f * (zk.y - zp.y));
        float2 w;
        if (sp) {
          w = sp[k];
        } else {
          float s, c;
          sincospif(-2.0f * (float)k / (float)Nf, &s, &c);
          w = make_float2(c, s);
        }
        const float2 wd = mx_cmul(w, d);
        X = make_float2(e.x + wd.y, e.y - wd.x);
        if (k == M) X = make_float2(Z[0].x - Z[0].y, 0.f);
      } else {
        X = Z[k];
      }
      row[k] = X;
      if (prow) prow[k] = fast_atan2f(X.y, X.x);
    }
    __syncthreads();       // the next frame overwrites the buffers these reads came from
  }
}

struct MixInvParams {
  const float2* X;
  const float* mag;
  const float* phase;
  const float* window;
  float* frames;  // (frames, n_fft)
  long long nframes;
  int n_fft, use_tw;
  MixPlan plan;
};

__global__ void irfft_mixed_kernel(MixInvParams p) {
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  const int Nf = p.n_fft, Fb = Nf / 2 + 1;
  const bool even = (Nf & 1) == 0;
  const int M = even ? Nf / 2 : Nf;
  float2* a = sm;
  float2* b = sm + M;       // M + 2 entries: until the FFT starts it stages the one-sided spectrum (even sizes)
  float2* tw = p.use_tw ? b + M + 2 : nullptr;
  float2* sp = (p.use_tw && even) ? tw + M : nullptr;
  if (tw) mx_fill_twiddles(tw, M, +1.0f);
  if (sp) mx_fill_split(sp, M, +1.0f);       // conj(W_N^k)
  const float sc = 1.0f / (float)Nf;
  for (long long f = blockIdx.x; f < p.nframes; f += gridDim.x) {
    auto bin = [&](int k) -> float2 {
      float2 v;
      if (p.X) {
        v = p.X[f * Fb + k];
      } else {
        float s, c;
        fast_sincosf(p.phase[f * Fb + k], s, c);       // as the register-core kernels: fp64 reduction + v_sin / v_cos
        const float m = p.mag[f * Fb + k];
        v = make_float2(m * c, m * s);
      }
      if (k == 0 || (even && k == M)) v.y = 0.f;     // a real signal: DC (and Nyquist) carry no imaginary part
      return v;
    };
    if (even) {
      float2* xs = b;
      for (int k = threadIdx.x; k <= M; k += blockDim.x) xs[k] = bin(k);
      __syncthreads();
      for (int k = threadIdx.x; k < M; k += blockDim.x) {
        const float2 xk = xs[k];
        float2 xp = xs[M - k];
        xp.y = -xp.y;
        const float2 e = mx_add(xk, xp);
        float2 w;
        if (sp) {
          w = sp[k];
        } else {
          float s, c;
          sincospif(2.0f * (float)k / (float)Nf, &s, &c);  // conj(W_N^k)
          w = make_float2(c, s);
        }
        const float2 d = mx_cmul(make_float2(xk.x - xp.x, xk.y - xp.y), w);
        a[k] = make_float2(e.x - d.y, e.y + d.x);
      }
    } else {
      for (int k = threadIdx.x; k < Fb; k += blockDim.x) {     // Hermitian extension of the one-sided spectrum
        const float2 v = bin(k);
        a[k] = v;
        if (k) a[Nf - k] = make_float2(v.x, -v.y);
      }
    }
    __syncthreads();
    const float2* z = mx_fft(a, b, M, +1.0f, tw, p.plan);
    float* dst = p.frames + f * Nf;
    if (even) {
      for (int n = threadIdx.x; n < M; n += blockDim.x) {
        dst[2 * n] = (z[n].x * sc) * p.window[2 * n];
        dst[2 * n + 1] = (z[n].y * sc) * p.window[2 * n + 1];
      }
    } else {
      for (int n = threadIdx.x; n < M; n += blockDim.x) dst[n] = (z[n].x * sc) * p.window[n];
    }
    __syncthreads();
  }
}

// radix plan of M: fours, then a two, then the odd primes in ascending order
static bool mix_plan(int M, MixPlan* plan) {
  plan->n_stages = 0;
  auto push = [&](int r) {
    if (plan->n_stages >= kMaxMixStages) return false;
    plan->radix[plan->n_stages++] = r;
    return true;
  };
  while (M % 4 == 0) { if (!push(4)) return false; M /= 4; }
  if (M % 2 == 0) { if (!push(2)) return false; M /= 2; }
  for (int p = 3; (long long)p * p <= M; p += 2)
    while (M % p == 0) { if (!push(p)) return false; M /= p; }
  if (M > 1 && !push(M)) return false;
  return true;
}

static int mix_set_lds(const void* fn, size_t bytes) {
  if (bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return -5;
  }
  return 0;
}

static int mix_threads(int M) {
  int t = (M / 4 + 63) / 64 * 64;
  return t < 64 ? 64 : (t > 256 ? 256 : t);
}

static unsigned mix_walkers(long long nframes) {
  const long long cap = 256LL * 16;
  return (unsigned)(nframes < cap ? nframes : cap);
}

int launch_rfft_mixed(const float* x, long long B, long long L, long long clip_stride, long long T, int n_fft, int hop,
                      int center, const float* window, float2* out, float* phase, hipStream_t stream) {
  if (B * T == 0) return 0;
  const int M = (n_fft & 1) ? n_fft : n_fft / 2;
  MixFwdParams p = {x, window, out, phase, B, L, clip_stride, T, n_fft, hop, center, M <= 4096, {}};
  if (!mix_plan(M, &p.plan)) return -2;
  const size_t lds = sizeof(float2) * (size_t)(2 * M + (p.use_tw ? 2 * M + 1 : 0));     // a, b (+ FFT and split twiddles)
  if (mix_set_lds((const void*)rfft_mixed_kernel, lds)) return -5;
  hipLaunchKernelGGL(rfft_mixed_kernel, dim3(mix_walkers(B * T)), dim3(mix_threads(M)), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft_mixed(const float2* X, const float* mag, const float* phase, long long nframes, int n_fft,
                       const float* window, float* frames, hipStream_t stream) {
  if (nframes == 0) return 0;
  const int M = (n_fft & 1) ? n_fft : n_fft / 2;
  MixInvParams p = {X, mag, phase, window, frames, nframes, n_fft, M <= 4096, {}};
  if (!mix_plan(M, &p.plan)) return -2;
  const size_t lds = sizeof(float2) * (size_t)(2 * M + 2 + (p.use_tw ? 2 * M + 1 : 0));
  if (mix_set_lds((const void*)irfft_mixed_kernel, lds)) return -5;
  hipLaunchKernelGGL(irfft_mixed_kernel, dim3(mix_walkers(nframes)), dim3(mix_threads(M)), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
