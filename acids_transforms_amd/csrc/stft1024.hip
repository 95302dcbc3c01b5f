// stft1024.hip -- fused framing + window + rFFT-1024 (forward) and
// irFFT-1024 + window + overlap-add (inverse) for gfx950.
//
// Replaces, for n_fft = 1024:
//   torch.stft(...).transpose(-2,-1)        reference transforms/stft.py:98-104, dgt.py:64-70   (K1)
//   x_fft.angle() phase buffer              stft.py:103, dgt.py:69                               (K2, optional)
//   torch.istft(...)                        stft.py:120-128, dgt.py:86-93                        (K3)
//   x * exp(1j*phase) before istft          stft.py:157-161, dgt.py:152-154                      (K15)
//   torch.fft.rfft(x*window) / irfft(x)*w   stft.py:249-266, dgt.py:285-302 (pre-framed)         (K4/K5)
//
// Execution model: one 64-lane wavefront owns one frame at a time and keeps
// window, twiddles (and, for the inverse, the overlap-add accumulators) in
// registers across a run of frames.  No workgroup barrier anywhere; a
// 256-thread block is just four independent waves sharing an LDS allocation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft512.h"

namespace at_hip {

constexpr int N = 1024;
constexpr int F = 513;
constexpr int WAVES_PER_BLOCK = 4;

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
struct FwdParams {
  const float* x;       // audio, clip b at x + b*clip_stride
  const float* window;  // N analysis window samples
  const float2* tw;     // twiddle table (fft512.h)
  float2* out;          // (B*T, 513) complex64
  float* phase;         // (B*T, 513) or nullptr
  long long B, L, clip_stride, T;
  long long total_frames;     // B*T
  long long frames_per_block; // multiple of WAVES_PER_BLOCK
  int hop;
  int center;  // 1: torch.stft center=True/reflect; 0: frame t starts at t*hop
};

__device__ __forceinline__ long long reflect_index(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// loads z[lane + 64 m] = (x[2n], x[2n+1]) of frame (b, t) into v
__device__ __forceinline__ void load_frame(const FwdParams& p, long long f, int lane, float2 (&v)[8]) {
  const long long b = f / p.T;
  const long long t = f - b * p.T;
  const float* clip = p.x + b * p.clip_stride;
  const long long start = t * (long long)p.hop - (p.center ? N / 2 : 0);
  const bool interior = (start >= 0) && (start + N <= p.L);
  const bool aligned = (((uintptr_t)(clip + start)) & 7) == 0;
  if (interior && aligned) {
    const float2* src = reinterpret_cast<const float2*>(clip + start);
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = src[lane + 64 * m];
  } else if (interior) {
    const float* src = clip + start;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      v[m].x = src[2 * (lane + 64 * m)];
      v[m].y = src[2 * (lane + 64 * m) + 1];
    }
  } else {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      long long i0 = start + 2 * (lane + 64 * m);
      if (p.center) {
        v[m].x = clip[reflect_index(i0, p.L)];
        v[m].y = clip[reflect_index(i0 + 1, p.L)];
      } else {  // zero padding past the end (utils/misc.py:156 pad())
        v[m].x = (i0 < p.L) ? clip[i0] : 0.0f;
        v[m].y = (i0 + 1 < p.L) ? clip[i0 + 1] : 0.0f;
      }
    }
  }
}

template <bool WRITE_PHASE>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void stft1024_fwd_kernel(FwdParams p) {
  __shared__ float2 lds_all[WAVES_PER_BLOCK * kFftLdsFloat2PerWave];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;

  Twiddles tw;
  load_twiddles<false>(tw, p.tw, lane);
  float2 win[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) win[m] = reinterpret_cast<const float2*>(p.window)[lane + 64 * m];

  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;

  long long f = f_begin + wave;
  float2 nxt[8];
  if (f < f_end) load_frame(p, f, lane, nxt);
  for (; f < f_end; f += WAVES_PER_BLOCK) {
    float2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = make_float2(nxt[m].x * win[m].x, nxt[m].y * win[m].y);
    // software prefetch of this wave's next frame
    if (f + WAVES_PER_BLOCK < f_end) load_frame(p, f + WAVES_PER_BLOCK, lane, nxt);

#if !defined(AT_ABLATE) || AT_ABLATE != 1
    fft512<false>(v, tw, lds, lane);
    float2 nyq;
    rfft_merge(v, tw, lane, nyq);
#else
    float2 nyq = v[0];
#endif

    float2* row = p.out + f * F;
#if defined(AT_ABLATE) && AT_ABLATE == 2
    if (v[0].x == 123456.789f && v[3].y == 2.5f)
#endif
    {
#pragma unroll
    for (int m = 0; m < 8; ++m) row[lane + 64 * m] = v[m];
    if (lane == 0) row[512] = nyq;
    }
    if (WRITE_PHASE) {
      float* prow = p.phase + f * F;
#pragma unroll
      for (int m = 0; m < 8; ++m) prow[lane + 64 * m] = atan2f(v[m].y, v[m].x);
      if (lane == 0) prow[512] = atan2f(nyq.y, nyq.x);
    }
  }
}

// ---------------------------------------------------------------------------
// forward, hop = 256 = N/4, center=True: sliding-window variant.
// A wave walks a run of consecutive frames of one clip.  Frame t+1 shares 768 of
// its 1024 samples with frame t, and in the lane layout z[l + 64 m] that overlap
// is exactly "registers m+2 of the same lane": the raw samples stay in registers,
// shifted by two slots per frame, and only the 256 new samples (two 8-byte loads
// per lane) are fetched per frame -- 1 KB of loads per frame instead of 4 KB.
// ---------------------------------------------------------------------------
// Banded filterbank for the fused |X| -> mel epilogue: filter n has its non-zero weights in
// rows [start[n], start[n]+len[n]) of the (F x N) bank, stored row-major in wT[n][0..lpad).
// slot[q*64 + lane] = filter handled by `lane` in pass q (-1 = none); long and short filters are paired.
struct BandBank {
  const int* start;
  const int* len;
  const int* slot;
  const float* wT;
  int n_filters, lpad, n_slots;
  int slot_len[4];   // longest band in each pass, rounded up to a multiple of 4 (<= lpad)
};
constexpr int kMaxBandFloats = 4096;   // LDS copy of the band weights (n_filters * lpad floats, 16 KB)

struct FwdRunParams {
  const float* x;
  const float* window;
  const float2* tw;
  float2* out;     // may be null when only the features are wanted (MEL == 2)
  float* phase;
  long long B, L, clip_stride, T;
  long long runs_per_clip, frames_per_run;
  // fused magnitude / mel epilogue (MEL != 0)
  BandBank bank;
  float* feat;            // (B*T, N) or, with feat_channel_major, (B, N, T)
  const float* offset;    // device scalars or null
  const float* scale;
  float eps;
  int contrast;           // 0 none, 1 log1p, 2 log, 3 log10
  int power2;             // |X|^2 instead of |X|
  int feat_channel_major;
};

// element n = lane + 64 m of the frame starting at padded position p0 (original index p0 - 512 + 2n)
__device__ __forceinline__ float2 load_pair(const float* clip, long long L, long long i0, bool interior, bool aligned,
                                            int lane_off) {
  // i0: original index of the first sample of this 64-lane, 128-sample segment; lane_off = 2*lane
  if (interior) {
    if (aligned) return reinterpret_cast<const float2*>(clip + i0)[lane_off >> 1];
    return make_float2(clip[i0 + lane_off], clip[i0 + lane_off + 1]);
  }
  return make_float2(clip[reflect_index(i0 + lane_off, L)], clip[reflect_index(i0 + lane_off + 1, L)]);
}

constexpr int FWD_WAVES = 4;  // 5 x 4.5 KB slabs + 15 KB tables = 37.7 KB -> 4 blocks = 20 waves per CU

__device__ __forceinline__ float fwd_contrast(float v, int mode, float eps) {
  switch (mode) {
    case 1: return logf(1.0f + v);
    case 2: return logf(fmaxf(v, eps));
    case 3: return log10f(fmaxf(v, eps));
    default: return v;
  }
}

// MEL: 0 = spectrum only; 1 = spectrum + fused banded-filterbank features; 2 = features only
template <bool WRITE_PHASE, int MEL>
__global__ __launch_bounds__(64 * FWD_WAVES, 3) void stft1024_h256_fwd_kernel(FwdRunParams p) {
  __shared__ float2 lds_all[FWD_WAVES * kFftLdsFloat2PerWave + kTwiddleCount + 512 + (MEL ? kMaxBandFloats / 2 : 0)];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + FWD_WAVES * kFftLdsFloat2PerWave;
  // workgroup-shared constants: twiddle table and analysis window
  for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * FWD_WAVES) tab[i] = p.tw[i];
  for (int i = threadIdx.x; i < 512; i += 64 * FWD_WAVES)
    tab[kTwiddleCount + i] = reinterpret_cast<const float2*>(p.window)[i];
  float* wlds = reinterpret_cast<float*>(tab + kTwiddleCount + 512);   // band weights, workgroup-shared
  if (MEL != 0)
    for (int i = threadIdx.x; i < p.bank.n_filters * p.bank.lpad; i += 64 * FWD_WAVES) wlds[i] = p.bank.wT[i];
  __syncthreads();

  const long long run = (long long)blockIdx.x * FWD_WAVES + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long t0 = r * p.frames_per_run;
  long long t1 = t0 + p.frames_per_run;
  if (t1 > p.T) t1 = p.T;
  if (t0 >= t1) return;

  Twiddles tw;
  load_twiddles<false>(tw, p.tw, lane);
  const float2* win = tab + kTwiddleCount;   // win[lane + 64 m]

  const float* clip = p.x + b * p.clip_stride;
  const bool clip_aligned = ((((uintptr_t)clip) & 7) == 0);  // frame starts are multiples of 256 samples
  const long long L = p.L;
  const int lane2 = 2 * lane;

  // first frame of the run: all eight segments
  float2 raw[8];
  {
    const long long s0 = t0 * 256 - 512;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const long long i0 = s0 + 128 * m;
      raw[m] = load_pair(clip, L, i0, i0 >= 0 && i0 + 128 <= L, clip_aligned, lane2);
    }
  }
  // Make the prologue loads architecturally complete here: otherwise the loop-top wait is the merge of
  // "just loaded" (from this prologue) and "loaded nine stores ago" (from the back edge) and collapses to
  // vmcnt(0), which would drain every frame's stores before the next frame starts.
#pragma unroll
  for (int m = 0; m < 8; ++m) asm volatile("" : "+v"(raw[m].x), "+v"(raw[m].y));
  float2* row = p.out + (b * p.T + t0) * F;
  float* prow = WRITE_PHASE ? p.phase + (b * p.T + t0) * F : nullptr;
  float* frow = (MEL != 0) ? p.feat + (b * p.T + t0) * (long long)p.bank.n_filters : nullptr;
  long long t_cur = t0;
  float mel_off = 0.f, mel_sc = 1.f;
  if (MEL != 0 && p.offset) {
    mel_off = *p.offset;
    mel_sc = *p.scale;
  }
  // this lane's filters (one per pass): index, first bank row, LDS offset of its weights -- loop invariant
  int mel_f[4] = {-1, -1, -1, -1}, mel_st[4] = {0, 0, 0, 0};
  if (MEL != 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < p.bank.n_slots) {
        mel_f[q] = p.bank.slot[q * 64 + lane];
        mel_st[q] = p.bank.start[mel_f[q] >= 0 ? mel_f[q] : 0];
      }
  }

  // one frame: window, FFT, merge, store; `n6`/`n7` are the next frame's two new segments, already requested.
  // The Nyquist bin (a one-lane, exec-masked store the compiler cannot count on) is deferred to the top of the
  // next iteration, *before* that iteration's loads: the wait for the prefetched samples then has exactly
  // eight younger stores behind it and leaves all of them in flight.
  float2 nyq_pending = make_float2(0.f, 0.f);
  float2* nyq_dst = nullptr;
  auto flush_nyquist = [&]() {
    if (nyq_dst != nullptr && lane == 0) *nyq_dst = nyq_pending;
    if (WRITE_PHASE && nyq_dst != nullptr && lane == 0)
      p.phase[(nyq_dst - p.out)] = atan2f(nyq_pending.y, nyq_pending.x);
  };
  auto frame_body = [&](float2 n6, float2 n7) {
    float2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float2 w = win[lane + 64 * m];
      v[m] = make_float2(raw[m].x * w.x, raw[m].y * w.y);
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) raw[m] = raw[m + 2];
    raw[6] = n6;
    raw[7] = n7;
    fft512<false>(v, tw, lds, lane);
    float2 nyq;
    rfft_merge(v, tw, lane, nyq);
    if (MEL != 2) {
#pragma unroll
      for (int m = 0; m < 8; ++m) row[lane + 64 * m] = v[m];
    }
    if (WRITE_PHASE) {
#pragma unroll
      for (int m = 0; m < 8; ++m) prow[lane + 64 * m] = atan2f(v[m].y, v[m].x);
      prow += F;
    }
    if (MEL != 0) {
      // |X| (or |X|^2) of this frame into the wave's LDS slab (free again after the FFT), then every lane
      // gathers the bands of its filters: sum_k |X[k]| w[k][n] over the band only (the rest of the column is 0)
      float* absrow = reinterpret_cast<float*>(lds);
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float s2 = fmaf(v[m].x, v[m].x, v[m].y * v[m].y);
        absrow[lane + 64 * m] = p.power2 ? s2 : __builtin_amdgcn_sqrtf(s2);
      }
      // bin 512, then zeros: a band walk may run up to lpad - 1 entries past its filter (zero weights there)
      absrow[512 + lane] = (lane == 0) ? (p.power2 ? nyq.x * nyq.x : fabsf(nyq.x)) : 0.0f;
      absrow[576 + lane] = 0.0f;
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= p.bank.n_slots) break;
        const int f = mel_f[q];
        const int fs = f >= 0 ? f : 0;
        // band starts and weight rows are 16-byte aligned: one ds_read_b128 each per four multiply-adds
        const float4* a = reinterpret_cast<const float4*>(absrow + mel_st[q]);
        const float4* w = reinterpret_cast<const float4*>(wlds + fs * p.bank.lpad);
        float acc = 0.f;
        const int quads = p.bank.slot_len[q] >> 2;       // wave-uniform; shorter bands multiply zeros
        for (int j = 0; j < quads; ++j) {
          const float4 av = a[j], wv = w[j];
          acc = fmaf(av.x, wv.x, acc);
          acc = fmaf(av.y, wv.y, acc);
          acc = fmaf(av.z, wv.z, acc);
          acc = fmaf(av.w, wv.w, acc);
        }
        if (f >= 0) {
          acc = fwd_contrast(acc, p.contrast, p.eps);
          if (p.offset) acc = (acc - mel_off) / mel_sc;
          if (p.feat_channel_major) p.feat[((long long)b * p.bank.n_filters + f) * p.T + t_cur] = acc;
          else frow[f] = acc;
        }
      }
      wave_lds_sync();
      frow += p.bank.n_filters;
      ++t_cur;
    }
    nyq_pending = nyq;
    nyq_dst = (MEL != 2) ? row + 512 : nullptr;
    row += F;
  };

  // Frames whose successor's new samples [256 t + 512, 256 t + 768) lie inside the clip take the
  // branch-free loop: two unconditional 8-byte loads issued *before* this frame's stores, so the
  // compiler's vmcnt accounting lets the stores stay in flight across iterations.
  long long t = t0;
  long long t_fast_end = (L >= 768) ? (L - 768) / 256 + 1 : 0;   // first t whose successor needs reflection
  if (t_fast_end > t1 - 1) t_fast_end = t1 - 1;        // the last frame of the run has no successor to fetch
  if (!clip_aligned) t_fast_end = t0;                  // odd-length clips: generic loop only
  if (t < t_fast_end) {
    const float2* nsrc = reinterpret_cast<const float2*>(clip + (t + 1) * 256 + 256) + lane;  // segment 6 of frame t+1
    for (; t < t_fast_end; ++t) {
      flush_nyquist();
      const float2 n6 = nsrc[0];
      const float2 n7 = nsrc[64];
      nsrc += 128;
      frame_body(n6, n7);
    }
  }
  for (; t < t1; ++t) {
    flush_nyquist();
    float2 n6 = make_float2(0.f, 0.f), n7 = n6;
    if (t + 1 < t1) {
      const long long i6 = (t + 1) * 256 - 512 + 768;
      n6 = load_pair(clip, L, i6, i6 >= 0 && i6 + 128 <= L, clip_aligned, lane2);
      n7 = load_pair(clip, L, i6 + 128, i6 + 128 >= 0 && i6 + 256 <= L, clip_aligned, lane2);
    }
    frame_body(n6, n7);
  }
  flush_nyquist();
}

// ---------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------
enum { IN_COMPLEX = 0, IN_POLAR = 1 };
enum { OUT_OLA = 0, OUT_FRAMES = 1 };

struct InvParams {
  const float2* X;      // (B*T, 513) complex64           (IN_COMPLEX)
  const float* mag;     // (B*T, 513)                      (IN_POLAR)
  const float* phase;   // (B*T, 513)                      (IN_POLAR)
  const float* window;  // N synthesis window samples
  const float* env;     // OUT_OLA: 16 x hop table, env[mask][r] = sum of window^2 over the frames in mask
  const float2* tw;
  float* y;             // OUT_OLA: (B, hop*(T-1));  OUT_FRAMES: (B*T, 1024)
  long long B, T;
  long long runs_per_clip;  // OUT_OLA
  long long slots_per_run;  // OUT_OLA
  long long total_frames, frames_per_block;  // OUT_FRAMES
};

// cos/sin of an unwrapped phase that may be ~1e5 rad (PGHI): reduce in fp64 to revolutions in
// [-0.5, 0.5] (exact to ~1e-16), then the hardware sin/cos (v_sin_f32 takes revolutions; abs error ~1e-6).
__device__ __forceinline__ void sincos_big(float phase, float& s, float& c) {
  double t = (double)phase * 0.15915494309189533577;  // 1 / (2 pi)
  t -= rint(t);
  const float r = (float)t;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

template <int IN_MODE>
__device__ __forceinline__ void load_spectrum(const InvParams& p, long long f, int lane, float2 (&v)[8], float& nyq_re) {
  if (IN_MODE == IN_COMPLEX) {
    const float2* row = p.X + f * F;
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = row[lane + 64 * m];
    nyq_re = row[512].x;  // broadcast load; only lane 0 uses it
  } else {
    const float* mrow = p.mag + f * F;
    const float* prow = p.phase + f * F;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float a = mrow[lane + 64 * m];
      float s, c;
      sincos_big(prow[lane + 64 * m], s, c);
      v[m] = make_float2(a * c, a * s);
    }
    float s, c;
    sincos_big(prow[512], s, c);
    nyq_re = mrow[512] * c;
  }
}

// one frame: spectrum -> windowed time samples y[m] = (x[2n], x[2n+1]) * w, n = lane + 64 m
template <typename TW>
__device__ __forceinline__ void synth_frame(float2 (&v)[8], float nyq_re, const TW& tw, const float2* win,
                                            float2* lds, int lane) {
  irfft_split(v, tw, lane, nyq_re);
  fft512<true>(v, tw, lds, lane);
  const float s = 1.0f / 1024.0f;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const float2 w = win[lane + 64 * m];   // synthesis window, workgroup-shared LDS copy
    v[m] = make_float2((v[m].x * s) * w.x, (v[m].y * s) * w.y);
  }
}

// K3: irFFT + window + overlap-add (hop = 256 = N/4) + envelope division + centre trim.
// A wave produces output hop-slots [j0, j1) of one clip, streaming over frames
// j0-1 .. j1+1 with the four overlapping frames' partial sums in registers.
template <int IN_MODE>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void istft1024_ola_kernel(InvParams p) {
  __shared__ float2 lds_all[WAVES_PER_BLOCK * kFftLdsFloat2PerWave + 512];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* win = lds_all + WAVES_PER_BLOCK * kFftLdsFloat2PerWave;
  for (int i = threadIdx.x; i < 512; i += 64 * WAVES_PER_BLOCK) win[i] = reinterpret_cast<const float2*>(p.window)[i];
  __syncthreads();

  const long long run = (long long)blockIdx.x * WAVES_PER_BLOCK + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long nslots = p.T - 1;
  const long long j0 = r * p.slots_per_run;
  long long j1 = j0 + p.slots_per_run;
  if (j1 > nslots) j1 = nslots;
  if (j0 >= j1) return;

  Twiddles tw;
  load_twiddles<true>(tw, p.tw, lane);

  float2 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = make_float2(0.f, 0.f);

  const long long fbase = b * p.T;
  float* yclip = p.y + b * (256 * nslots);
  long long t = (j0 > 0) ? j0 - 1 : 0;
  const long long t_last = j1 + 1;  // inclusive; frames >= T contribute nothing

  float2 nxt[8];
  float nxt_nyq = 0.f;
  if (t < p.T) load_spectrum<IN_MODE>(p, fbase + t, lane, nxt, nxt_nyq);
  // accumulators start aligned with frame t: acc[m] covers padded samples t*256 + 2*(lane+64m)
  for (; t <= t_last; ++t) {
    if (t < p.T) {
      float2 v[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = nxt[m];
      float nyq = nxt_nyq;
      if (t + 1 <= t_last && t + 1 < p.T) load_spectrum<IN_MODE>(p, fbase + t + 1, lane, nxt, nxt_nyq);
      synth_frame(v, nyq, tw, win, lds, lane);
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = cadd(acc[m], v[m]);
    }
    const long long j = t - 2;  // padded slot t is complete -> output slot j
    if (j >= j0 && j < j1) {
      // frames contributing to output slot j are j-1 .. j+2
      int mask = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        long long tt = j - 1 + q;
        if (tt >= 0 && tt < p.T) mask |= 1 << q;
      }
      const float2* env = reinterpret_cast<const float2*>(p.env + mask * 256);
      float2* dst = reinterpret_cast<float2*>(yclip + j * 256);
      float2 e0 = env[lane], e1 = env[lane + 64];
      dst[lane] = make_float2(acc[0].x / e0.x, acc[0].y / e0.y);
      dst[lane + 64] = make_float2(acc[1].x / e1.x, acc[1].y / e1.y);
    }
    // advance the accumulator window by one hop (= 2 register slots)
#pragma unroll
    for (int m = 0; m < 6; ++m) acc[m] = acc[m + 2];
    acc[6] = make_float2(0.f, 0.f);
    acc[7] = make_float2(0.f, 0.f);
  }
}

// K5: irFFT + window, frames out (no overlap-add): RealtimeSTFT/RealtimeDGT.invert
template <int IN_MODE>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void irfft1024_frames_kernel(InvParams p) {
  __shared__ float2 lds_all[WAVES_PER_BLOCK * kFftLdsFloat2PerWave + 512];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* win = lds_all + WAVES_PER_BLOCK * kFftLdsFloat2PerWave;
  for (int i = threadIdx.x; i < 512; i += 64 * WAVES_PER_BLOCK) win[i] = reinterpret_cast<const float2*>(p.window)[i];
  __syncthreads();
  Twiddles tw;
  load_twiddles<true>(tw, p.tw, lane);
  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;
  for (long long f = f_begin + wave; f < f_end; f += WAVES_PER_BLOCK) {
    float2 v[8];
    float nyq;
    load_spectrum<IN_MODE>(p, f, lane, v, nyq);
    synth_frame(v, nyq, tw, win, lds, lane);
    float2* dst = reinterpret_cast<float2*>(p.y + f * N);
#pragma unroll
    for (int m = 0; m < 8; ++m) dst[lane + 64 * m] = v[m];
  }
}

}  // namespace at_hip

// ---------------------------------------------------------------------------
// host launchers (C++ linkage inside the library; the extern "C" ABI is capi.hip)
// ---------------------------------------------------------------------------
namespace at_hip {

static inline int num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

int launch_stft1024_fwd(const float* x, long long B, long long L, long long clip_stride, long long T, int hop,
                        int center, const float* window, const float2* tw, float2* out, float* phase,
                        hipStream_t stream) {
  FwdParams p;
  p.x = x; p.window = window; p.tw = tw; p.out = out; p.phase = phase;
  p.B = B; p.L = L; p.clip_stride = clip_stride; p.T = T; p.hop = hop; p.center = center;
  p.total_frames = B * T;
  if (p.total_frames == 0) return 0;
  // persistent-ish grid: up to 4 blocks (16 waves) per CU, each block a contiguous run of frames
  long long max_blocks = (long long)num_cus() * 4;
  long long fpb = (p.total_frames + max_blocks - 1) / max_blocks;
  fpb = ((fpb + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK) * WAVES_PER_BLOCK;
  if (fpb < 2 * WAVES_PER_BLOCK) fpb = 2 * WAVES_PER_BLOCK;
  p.frames_per_block = fpb;
  long long blocks = (p.total_frames + fpb - 1) / fpb;
  if (phase)
    hipLaunchKernelGGL(stft1024_fwd_kernel<true>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  else
    hipLaunchKernelGGL(stft1024_fwd_kernel<false>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_stft1024_h256_fwd(const float* x, long long B, long long L, long long clip_stride, long long T,
                             const float* window, const float2* tw, float2* out, float* phase, const BandBank* bank,
                             float* feat, const float* offset, const float* scale, float eps, int contrast, int power2,
                             int feat_channel_major, hipStream_t stream) {
  FwdRunParams p = {};
  p.x = x; p.window = window; p.tw = tw; p.out = out; p.phase = phase;
  p.B = B; p.L = L; p.clip_stride = clip_stride; p.T = T;
  if (bank) {
    p.bank = *bank; p.feat = feat; p.offset = offset; p.scale = scale; p.eps = eps; p.contrast = contrast;
    p.power2 = power2; p.feat_channel_major = feat_channel_major;
  }
  if (B * T == 0) return 0;
  // ~16 waves per CU; runs of at least 24 frames so that the 3 extra segment loads of a run start stay < 5 %
  long long target_waves = (long long)num_cus() * 16;
  long long runs_per_clip = (target_waves + B - 1) / B;
  if (runs_per_clip < 1) runs_per_clip = 1;
  long long fpr = (T + runs_per_clip - 1) / runs_per_clip;
  if (fpr < 24) fpr = 24;
  if (fpr > T) fpr = T;
  runs_per_clip = (T + fpr - 1) / fpr;
  p.runs_per_clip = runs_per_clip;
  p.frames_per_run = fpr;
  long long waves = B * runs_per_clip;
  long long blocks = (waves + FWD_WAVES - 1) / FWD_WAVES;
  const dim3 grid((unsigned)blocks), block(64 * FWD_WAVES);
  if (!bank) {
    if (phase) hipLaunchKernelGGL((stft1024_h256_fwd_kernel<true, 0>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((stft1024_h256_fwd_kernel<false, 0>), grid, block, 0, stream, p);
  } else if (out) {
    if (phase) hipLaunchKernelGGL((stft1024_h256_fwd_kernel<true, 1>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((stft1024_h256_fwd_kernel<false, 1>), grid, block, 0, stream, p);
  } else {
    hipLaunchKernelGGL((stft1024_h256_fwd_kernel<false, 2>), grid, block, 0, stream, p);
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_istft1024_ola(const float2* X, const float* mag, const float* phase, long long B, long long T,
                         const float* window, const float* env16, const float2* tw, float* y, hipStream_t stream) {
  InvParams p = {};
  p.X = X; p.mag = mag; p.phase = phase; p.window = window; p.env = env16; p.tw = tw; p.y = y;
  p.B = B; p.T = T;
  const long long nslots = T - 1;
  if (B == 0 || nslots <= 0) return 0;
  // choose the run length so that the grid has ~12 waves per CU but runs are >= 32 slots
  long long target_waves = (long long)num_cus() * 12;
  long long runs_per_clip = (target_waves + B - 1) / B;
  if (runs_per_clip < 1) runs_per_clip = 1;
  long long spr = (nslots + runs_per_clip - 1) / runs_per_clip;
  if (spr < 32) spr = 32;
  if (spr > nslots) spr = nslots;
  runs_per_clip = (nslots + spr - 1) / spr;
  p.runs_per_clip = runs_per_clip;
  p.slots_per_run = spr;
  long long waves = B * runs_per_clip;
  long long blocks = (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (X)
    hipLaunchKernelGGL(istft1024_ola_kernel<IN_COMPLEX>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  else
    hipLaunchKernelGGL(istft1024_ola_kernel<IN_POLAR>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft1024_frames(const float2* X, const float* mag, const float* phase, long long nframes,
                            const float* window, const float2* tw, float* y, hipStream_t stream) {
  InvParams p = {};
  p.X = X; p.mag = mag; p.phase = phase; p.window = window; p.tw = tw; p.y = y;
  p.total_frames = nframes;
  if (nframes == 0) return 0;
  long long max_blocks = (long long)num_cus() * 4;
  long long fpb = (nframes + max_blocks - 1) / max_blocks;
  fpb = ((fpb + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK) * WAVES_PER_BLOCK;
  p.frames_per_block = fpb;
  long long blocks = (nframes + fpb - 1) / fpb;
  if (X)
    hipLaunchKernelGGL(irfft1024_frames_kernel<IN_COMPLEX>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  else
    hipLaunchKernelGGL(irfft1024_frames_kernel<IN_POLAR>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
