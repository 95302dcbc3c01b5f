// stft_generic.hip -- the same transforms as stft1024.hip for any power-of-two
// n_fft in [8, 16384] and any hop (the reference accepts arbitrary sizes:
// transforms/stft.py:67-75).  A workgroup per frame at a time (it walks frames blockIdx.x + k gridDim.x with its twiddle
// tables filled once), radix-4 / radix-2 Stockham in LDS.
// Correctness path for the non-default sizes the parity tests use; the
// n_fft = 1024 kernels in stft1024.hip are the tuned ones.
#include <hip/hip_runtime.h>
#include "fastmath.h"
#include <stdint.h>

namespace at_hip {

struct GenFwdParams {
  const float* x;
  const float* window;
  float2* out;
  float* phase;
  long long B, L, clip_stride, T;
  int n_fft, hop, center;
  int use_tw;   // the twiddle table fits in LDS
};

__device__ __forceinline__ long long g_reflect(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

__device__ __forceinline__ float2 g_cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// tw[j] = exp(sign * 2 pi i j / M), j < M / 2: every stage's twiddle exp(sign * i pi k / Ns) is entry k * M / (2 Ns) --
// the same sincospif argument (k / Ns = 2 j / M exactly), evaluated once per workgroup instead of once per butterfly.
__device__ void fill_twiddles(float2* tw, int M, float sign) {
  for (int j = threadIdx.x; j < 3 * M / 4 + 1; j += blockDim.x) {       // radix-4 stages reach 3 k M / (4 Ns) < 3 M / 4
    float s, c;
    sincospif(sign * 2.0f * (float)j / (float)M, &s, &c);
    tw[j] = make_float2(c, s);
  }
}

// Stockham autosort FFT of M points held in `a` (ping) with scratch `b`.  Returns the pointer that holds the result.
// sign = -1 forward, +1 inverse.  tw: fill_twiddles table (radix-4 stages, one radix-2 stage first when log2 M is
// odd: half the LDS passes and barriers) or null (n_fft > 4096: radix-2 with sincospi per butterfly).
__device__ float2* stockham(float2* a, float2* b, int M, float sign, const float2* tw) {
  if (tw && M >= 4) {
    int Ns = 1;
    if (__popc(M - 1) & 1) {          // log2 M odd: one radix-2 stage (Ns = 1: all twiddles are 1)
      for (int j = threadIdx.x; j < M / 2; j += blockDim.x) {
        const float2 u = a[j], v = a[j + M / 2];
        b[2 * j] = make_float2(u.x + v.x, u.y + v.y);
        b[2 * j + 1] = make_float2(u.x - v.x, u.y - v.y);
      }
      __syncthreads();
      float2* t = a; a = b; b = t;
      Ns = 2;
    }
    for (; Ns < M; Ns <<= 2) {
      const int tstep = (M / 4) / Ns;        // twiddle exp(sign 2 pi i r k / (4 Ns)) = tw[r k tstep]
      for (int j = threadIdx.x; j < M / 4; j += blockDim.x) {
        const int k = j & (Ns - 1);
        const float2 v0 = a[j];
        const float2 v1 = g_cmul(a[j + M / 4], tw[k * tstep]);
        const float2 v2 = g_cmul(a[j + M / 2], tw[2 * k * tstep]);
        const float2 v3 = g_cmul(a[j + 3 * M / 4], tw[3 * k * tstep]);
        const float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y), d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
        const float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y), d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
        // sign * i * d13
        const float2 r = make_float2(-sign * d13.y, sign * d13.x);
        const int j0 = ((j - k) << 2) + k;
        b[j0] = make_float2(s02.x + s13.x, s02.y + s13.y);
        b[j0 + Ns] = make_float2(d02.x + r.x, d02.y + r.y);
        b[j0 + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
        b[j0 + 3 * Ns] = make_float2(d02.x - r.x, d02.y - r.y);
      }
      __syncthreads();
      float2* t = a; a = b; b = t;
    }
    return a;
  }
  for (int Ns = 1; Ns < M; Ns <<= 1) {
    const int tstep = (M / 2) / Ns;
    for (int j = threadIdx.x; j < M / 2; j += blockDim.x) {
      int k = j & (Ns - 1);
      float s, c;
      if (tw) {
        const float2 w = tw[k * tstep];
        c = w.x;
        s = w.y;
      } else {
        sincospif(sign * (float)k / (float)Ns, &s, &c);  // angle = sign * 2*pi*k/(2 Ns)
      }
      float2 u = a[j];
      float2 v = g_cmul(a[j + M / 2], make_float2(c, s));
      int j0 = ((j - k) << 1) + k;
      b[j0] = make_float2(u.x + v.x, u.y + v.y);
      b[j0 + Ns] = make_float2(u.x - v.x, u.y - v.y);
    }
    __syncthreads();
    float2* t = a;
    a = b;
    b = t;
  }
  return a;
}

// exp(sign * pi i k / M) = exp(sign * 2 pi i k / n_fft), k = 0 .. M: the twiddle of the real split, once per workgroup
__device__ void fill_split_twiddles(float2* sp, int M, float sign) {
  for (int k = threadIdx.x; k <= M; k += blockDim.x) {
    float s, c;
    sincospif(sign * (float)k / (float)M, &s, &c);
    sp[k] = make_float2(c, s);
  }
}

// A workgroup walks frames blockIdx.x, blockIdx.x + gridDim.x, ...: the twiddle tables are filled once per workgroup,
// not once per frame (their sincospi calls used to outnumber the butterflies' own arithmetic).
__global__ void rfft_generic_kernel(GenFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  const int Nf = p.n_fft, M = Nf / 2;
  float2* a = sm;
  float2* b = sm + M;
  float2* tw = p.use_tw ? sm + 2 * M : nullptr;
  float2* sp = p.use_tw ? tw + (3 * M / 4 + 1) : nullptr;
  if (tw) {
    fill_twiddles(tw, M, -1.0f);
    fill_split_twiddles(sp, M, -1.0f);
  }
  const int Fb = M + 1;
  for (long long f = blockIdx.x; f < p.B * p.T; f += gridDim.x) {
    const long long bidx = f / p.T, t = f - bidx * p.T;
    const float* clip = p.x + bidx * p.clip_stride;
    const long long start = t * (long long)p.hop - (p.center ? Nf / 2 : 0);
    for (int n = threadIdx.x; n < M; n += blockDim.x) {
      long long i0 = start + 2 * n, i1 = i0 + 1;
      float x0, x1;
      if (p.center) {
        x0 = clip[g_reflect(i0, p.L)];
        x1 = clip[g_reflect(i1, p.L)];
      } else {
        x0 = (i0 < p.L) ? clip[i0] : 0.f;
        x1 = (i1 < p.L) ? clip[i1] : 0.f;
      }
      a[n] = make_float2(x0 * p.window[2 * n], x1 * p.window[2 * n + 1]);
    }
    __syncthreads();
    float2* Z = stockham(a, b, M, -1.0f, tw);
    float2* row = p.out + f * Fb;
    float* prow = p.phase ? p.phase + f * Fb : nullptr;
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
      float2 zk = Z[k & (M - 1)];
      float2 zp = Z[(M - k) & (M - 1)];
      zp.y = -zp.y;
      float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y + zp.y));
      float2 d = make_float2(0.5f * (zk.x - zp.x), 0.5f * (zk.y - zp.y));
      float2 w;
      if (sp) {
        w = sp[k];
      } else {
        float s, c;
        sincospif(-2.0f * (float)k / (float)Nf, &s, &c);
        w = make_float2(c, s);
      }
      float2 wd = g_cmul(w, d);
      float2 X = make_float2(e.x + wd.y, e.y - wd.x);
      if (k == M) X = make_float2(Z[0].x - Z[0].y, 0.f);
      row[k] = X;
      if (prow) prow[k] = fast_atan2f(X.y, X.x);
    }
    __syncthreads();       // the next frame overwrites the buffers these reads came from
  }
}

struct GenInvParams {
  const float2* X;
  const float* mag;
  const float* phase;
  const float* window;
  float* frames;  // (B*T, n_fft)
  long long nframes;
  int n_fft;
  int use_tw;
};

__global__ void irfft_generic_kernel(GenInvParams p) {
  extern __shared__ __attribute__((aligned(16))) float2 sm[];
  const int Nf = p.n_fft, M = Nf / 2, Fb = M + 1;
  float2* a = sm;
  float2* b = sm + M;       // M + 2 entries: until the FFT starts it stages the one-sided spectrum (Fb = M + 1)
  float2* xs = b;
  float2* tw = p.use_tw ? b + M + 2 : nullptr;
  float2* sp = p.use_tw ? tw + (3 * M / 4 + 1) : nullptr;
  if (tw) {
    fill_twiddles(tw, M, +1.0f);
    fill_split_twiddles(sp, M, +1.0f);     // conj(W_N^k)
  }
  const float sc = 1.0f / (float)Nf;
  for (long long f = blockIdx.x; f < p.nframes; f += gridDim.x) {
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
      float2 v;
      if (p.X) {
        v = p.X[f * Fb + k];
      } else {
        float s, c;
        fast_sincosf(p.phase[f * Fb + k], s, c);       // as the register-core kernels: fp64 reduction + v_sin / v_cos
        float m = p.mag[f * Fb + k];
        v = make_float2(m * c, m * s);
      }
      if (k == 0 || k == M) v.y = 0.f;
      xs[k] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
      float2 xk = xs[k];
      float2 xp = xs[M - k];
      xp.y = -xp.y;
      float2 e = make_float2(xk.x + xp.x, xk.y + xp.y);
      float2 w;
      if (sp) {
        w = sp[k];
      } else {
        float s, c;
        sincospif(2.0f * (float)k / (float)Nf, &s, &c);  // conj(W_N^k)
        w = make_float2(c, s);
      }
      float2 d = g_cmul(make_float2(xk.x - xp.x, xk.y - xp.y), w);
      a[k] = make_float2(e.x - d.y, e.y + d.x);
    }
    __syncthreads();
    float2* z = stockham(a, b, M, +1.0f, tw);
    float2* dst = reinterpret_cast<float2*>(p.frames + f * Nf);
    for (int n = threadIdx.x; n < M; n += blockDim.x)
      dst[n] = make_float2((z[n].x * sc) * p.window[2 * n], (z[n].y * sc) * p.window[2 * n + 1]);
    __syncthreads();
  }
}

struct OlaParams {
  const float* frames;  // (B, T, n_fft)
  const float* window;
  float* y;             // (B, hop*(T-1))
  long long B, T;
  int n_fft, hop;
};

// gather-form overlap-add with torch.istft's envelope division and centre trim
__global__ void ola_gather_kernel(OlaParams p) {
  const long long out_len = (long long)p.hop * (p.T - 1) + (p.n_fft & 1);   // torch.istft trims n_fft / 2 (floor) at both ends
  const long long total = p.B * out_len;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / out_len, s = i - b * out_len;
    const long long pp = s + p.n_fft / 2;  // position in the padded signal
    long long t_hi = pp / p.hop;
    if (t_hi > p.T - 1) t_hi = p.T - 1;
    long long t_lo = (pp - p.n_fft + p.hop) / p.hop;  // ceil((pp - n_fft + 1)/hop)
    if (pp - p.n_fft + 1 <= 0) t_lo = 0;
    float acc = 0.f, env = 0.f;
    for (long long t = t_lo; t <= t_hi; ++t) {
      const int o = (int)(pp - t * p.hop);
      if (o < 0 || o >= p.n_fft) continue;
      acc += p.frames[(b * p.T + t) * p.n_fft + o];
      const float w = p.window[o];
      env += w * w;
    }
    p.y[i] = acc / env;
  }
}

// the same four samples at a time (hop, n_fft multiples of 4, 16-byte aligned buffers): an aligned group of four
// padded positions never straddles a frame start, so all four samples see the same frames t_lo .. t_hi and the same
// summation order as the scalar kernel
__global__ void ola_gather4_kernel(OlaParams p) {
  const long long out_len = (long long)p.hop * (p.T - 1);
  const long long total4 = p.B * out_len / 4;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4;
       i4 += (long long)gridDim.x * blockDim.x) {
    const long long i = 4 * i4;
    const long long b = i / out_len, s = i - b * out_len;
    const long long pp = s + p.n_fft / 2;
    long long t_hi = pp / p.hop;
    if (t_hi > p.T - 1) t_hi = p.T - 1;
    long long t_lo = (pp - p.n_fft + p.hop) / p.hop;
    if (pp - p.n_fft + 1 <= 0) t_lo = 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), env = acc;
    for (long long t = t_lo; t <= t_hi; ++t) {
      const int o = (int)(pp - t * p.hop);
      if (o < 0 || o >= p.n_fft) continue;
      const float4 v = *reinterpret_cast<const float4*>(p.frames + (b * p.T + t) * p.n_fft + o);
      const float4 w = *reinterpret_cast<const float4*>(p.window + o);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      env.x += w.x * w.x; env.y += w.y * w.y; env.z += w.z * w.z; env.w += w.w * w.w;
    }
    *reinterpret_cast<float4*>(p.y + i) = make_float4(acc.x / env.x, acc.y / env.y, acc.z / env.z, acc.w / env.w);
  }
}

static int set_lds(const void* fn, size_t bytes) {
  if (bytes > 64 * 1024) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      (void)hipGetLastError();   // do not leave the error latched for the next launch's check
      return -5;
    }
  }
  return 0;
}

// workgroups of a launch: every frame its own while there are few, a few resident rounds of frame walkers above that
static unsigned frame_walkers(long long nframes) {
  const long long cap = 256LL * 16;
  return (unsigned)(nframes < cap ? nframes : cap);
}

int launch_rfft_generic(const float* x, long long B, long long L, long long clip_stride, long long T, int n_fft,
                        int hop, int center, const float* window, float2* out, float* phase, hipStream_t stream) {
  if (B * T == 0) return 0;
  const int use_tw = n_fft <= 8192;      // 16384: the tables do not fit next to the two frame buffers
  GenFwdParams p = {x, window, out, phase, B, L, clip_stride, T, n_fft, hop, center, use_tw};
  // 2 M (+ 3 M / 4 + 1 FFT twiddles + M + 1 split twiddles)
  size_t lds = sizeof(float2) * (size_t)(n_fft + (use_tw ? 3 * n_fft / 8 + 1 + n_fft / 2 + 1 : 0));
  if (set_lds((const void*)rfft_generic_kernel, lds)) return -5;
  int threads = n_fft / 4 < 64 ? 64 : (n_fft / 4 > 256 ? 256 : n_fft / 4);
  hipLaunchKernelGGL(rfft_generic_kernel, dim3(frame_walkers(B * T)), dim3(threads), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft_generic(const float2* X, const float* mag, const float* phase, long long nframes, int n_fft,
                         const float* window, float* frames, hipStream_t stream) {
  if (nframes == 0) return 0;
  const int use_tw = n_fft <= 8192;      // 16384: the tables do not fit next to the two frame buffers
  GenInvParams p = {X, mag, phase, window, frames, nframes, n_fft, use_tw};
  size_t lds = sizeof(float2) * (size_t)(n_fft + 2 + (use_tw ? 3 * n_fft / 8 + 1 + n_fft / 2 + 1 : 0));   // a, b (+ twiddles)
  if (set_lds((const void*)irfft_generic_kernel, lds)) return -5;
  int threads = n_fft / 4 < 64 ? 64 : (n_fft / 4 > 256 ? 256 : n_fft / 4);
  hipLaunchKernelGGL(irfft_generic_kernel, dim3(frame_walkers(nframes)), dim3(threads), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_ola_gather(const float* frames, long long B, long long T, int n_fft, int hop, const float* window,
                      float* y, hipStream_t stream) {
  long long total = B * ((long long)hop * (T - 1) + (n_fft & 1));
  if (total <= 0) return 0;
  OlaParams p = {frames, window, y, B, T, n_fft, hop};
  const bool vec4 = (hop % 4 == 0) && (n_fft % 8 == 0) && (((uintptr_t)frames) & 15) == 0 && (((uintptr_t)window) & 15) == 0 &&
                    (((uintptr_t)y) & 15) == 0;
  long long blocks = ((vec4 ? total / 4 : total) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (vec4) hipLaunchKernelGGL(ola_gather4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(ola_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
