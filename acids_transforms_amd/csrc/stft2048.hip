// stft2048.hip -- n_fft = 2048 on the one-wavefront register FFT core (fft512.h).
//
// Replaces, for n_fft = 2048 (any hop):
//   torch.stft(...).transpose(-2,-1)          reference transforms/stft.py:98-104, dgt.py:64-70
//   torch.fft.rfft(x*window) on frames        stft.py:249-253, dgt.py:285-289
//   torch.fft.irfft(X) * inv_window           stft.py:260-266, dgt.py:296-302; the frames of torch.istft
//                                             (stft.py:120-128), overlap-added by stft_generic.hip's gather
// Until round 2 this size ran on the workgroup-per-frame LDS Stockham kernel of stft_generic.hip (~1.4 TB/s).
//
// A 2048-point real transform is a 1024-point complex FFT of z[n] = x[2n] + i x[2n+1] plus the real split; the
// 1024-point FFT is two 512-point FFTs (the wave-level radix-8 core: 8 complex points per lane) of the even and the
// odd complex samples plus one radix-2 stage:
//   ze[m] = z[2m]   = x[4m]   + i x[4m+1]        -> one float4 load per lane and register holds (ze, zo) of the
//   zo[m] = z[2m+1] = x[4m+2] + i x[4m+3]           same m: 1 KB contiguous per wave instruction
//   Z[k] = Ze[k] + W1024^k Zo[k],  Z[k+512] = Ze[k] - W1024^k Zo[k]     (k = lane + 64 m: lane-local)
//   X[k] = (Z[k] + conj Z[1024-k])/2 - (i/2) W2048^k (Z[k] - conj Z[1024-k]),  k = 0 .. 1024
// The mirror partner Z[1024-k] lives in lane 64-lane, register 15-m (lane 0: its own register 16-m).  The inverse
// runs the same steps backwards.  One wave = one frame at a time, frames of a block interleaved over its four
// waves, the next frame's loads issued before the current frame's FFTs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"
#include "fastmath.h"
#include "fft512.h"
#include "band_bank.h"
#include "mel_gemm.h"   // C_* contrast codes
#include "run_plan.h"
#include "variants.h"
#include <stdlib.h>

namespace at_hip {

constexpr int N2K = 2048;
constexpr int F2K = N2K / 2 + 1;     // 1025
constexpr int W2K = 4;               // waves per block

struct P2k {
  const float* x;        // forward: audio, clip b at x + b*clip_stride
  const float* window;   // 2048 samples (analysis or synthesis)
  const float2* tw;      // fft512 twiddle table (capi.hip)
  const float2* tw2k;    // W2048^k, k = 0 .. 1023
  float2* X;             // (frames, 1025) complex64: forward output / inverse input
  const float* mag;      // inverse, polar input
  const float* phase;
  float* phase_out;      // forward: optional angle output (frames, 1025)
  float* y;              // inverse: (frames, 2048) windowed time frames
  long long L, clip_stride, T, total_frames, frames_per_block;
  int hop, center;
};

__device__ __forceinline__ long long reflect2k(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// q[j] = x[s + 4 (lane + 64 j) .. + 3] of frame f (reflect padding with center, zero padding without)
__device__ __forceinline__ void load_frame2k(const P2k& p, long long f, int lane, float4 (&q)[8]) {
  const long long b = f / p.T, t = f - b * p.T;
  const float* clip = p.x + b * p.clip_stride;
  const long long start = t * (long long)p.hop - (p.center ? N2K / 2 : 0);
  const bool interior = (start >= 0) && (start + N2K <= p.L);
  if (interior && ((((uintptr_t)(clip + start)) & 15) == 0)) {
    const float4* src = reinterpret_cast<const float4*>(clip + start);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = src[lane + 64 * j];
  } else if (interior) {
    const float* src = clip + start;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 4 * (lane + 64 * j);
      q[j] = make_float4(src[i], src[i + 1], src[i + 2], src[i + 3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long i0 = start + 4 * (lane + 64 * j);
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long long i = i0 + c;
        if (p.center) v[c] = clip[reflect2k(i, p.L)];
        else v[c] = (i >= 0 && i < p.L) ? clip[i] : 0.0f;     // zero padding past the end (utils/misc.py:156)
      }
      q[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// mirror partners P[m] = Z[(1024 - (lane + 64 m)) mod 1024] of the 16 registers
__device__ __forceinline__ void mirror1024(const v2f (&v)[16], v2f (&p)[16], int lane) {
  const int src = (64 - lane) & 63;
  v2f q[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const v2f a = q[15 - m];            // lane > 0: lane 64 - lane, register 15 - m
    const v2f b = q[(16 - m) & 15];     // lane 0: own register (16 - m) mod 16
    p[m] = (lane == 0) ? b : a;
  }
}

// Both kernels keep the constant tables in LDS, shared by the block's four waves and read at the point of use (the
// fft512 twiddles: 11 KB; W2048^k: 8 KB): held in registers they cost 76 VGPRs and leave one wave per SIMD.
template <bool INV>
__device__ __forceinline__ void stage_tables2k(const P2k& p, float2* tab, float2* w2tab) {
  for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * W2K) tab[i] = twiddle_for_lds<INV>(p.tw, i);
  for (int i = threadIdx.x; i < 1024; i += 64 * W2K) w2tab[i] = p.tw2k[i];
  __syncthreads();
}

template <bool WRITE_PHASE>
__global__ __launch_bounds__(64 * W2K, 3) void stft2048_fwd_kernel(P2k p) {
  __shared__ float2 lds_all[W2K * kFftLdsFloat2PerWave + kTwiddleCount + 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W2K * kFftLdsFloat2PerWave;
  const v2f* w2 = reinterpret_cast<const v2f*>(tab + kTwiddleCount) + lane;     // w2[64 m] = W2048^(lane + 64 m)
  stage_tables2k<false>(p, tab, tab + kTwiddleCount);
  const LdsTwiddles<false> tw = {tab, lane};     // tw.getr(m) = W1024^(lane + 64 m) / 2
  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;
  const float4* win4 = reinterpret_cast<const float4*>(p.window);
  const v2f hh = {0.5f, 0.5f};

  long long f = f_begin + wave;
  float4 nxt[8];
  if (f < f_end) load_frame2k(p, f, lane, nxt);
  for (; f < f_end; f += W2K) {
    v2f ze[8], zo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 w = win4[lane + 64 * j];   // L1/L2-resident (8 KB), re-read per frame instead of 32 VGPRs
      ze[j] = (v2f){nxt[j].x * w.x, nxt[j].y * w.y};
      zo[j] = (v2f){nxt[j].z * w.z, nxt[j].w * w.w};
    }
    if (f + W2K < f_end) load_frame2k(p, f + W2K, lane, nxt);
    fft512<false>(ze, tw, lds, lane);
    fft512<false>(zo, tw, lds, lane);
    // radix-2: H = Z / 2 (the real split wants the half): H[k] = Ze/2 + (W1024^k / 2) Zo, H[k+512] = Ze/2 - ...
    v2f z[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v2f t = cmul_v(zo[m], tw.getr(m));
      const v2f e = ze[m] * hh;
      z[m] = e + t;
      z[m + 8] = e - t;
    }
    v2f pm[16];
    mirror1024(z, pm, lane);
    // X[k] = (H[k] + conj H') - i W2048^k (H[k] - conj H'),  H = Z / 2,  H' = H[1024 - k]  (k = 0: H' = H[0], X[0] real)
    const float2 nyq = make_float2(2.0f * (z[0].x - z[0].y), 0.0f);    // X[1024] = Re Z[0] - Im Z[0] (lane 0)
    float2* row = p.X + f * F2K;
    float* prow = WRITE_PHASE ? p.phase_out + f * F2K : nullptr;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const v2f e = add_conj(z[m], pm[m]);
      const v2f d = sub_conj(z[m], pm[m]);
      const v2f xk = add_mi(e, cmul_v(d, w2[64 * m]));      // e - i W d
      row[lane + 64 * m] = to_f2(xk);
      if (WRITE_PHASE) prow[lane + 64 * m] = fast_atan2f(xk.y, xk.x);
    }
    if (lane == 0) {
      row[1024] = nyq;
      if (WRITE_PHASE) prow[1024] = fast_atan2f(nyq.y, nyq.x);
    }
  }
}


// ---------------------------------------------------------------------------
// forward, hop = 512 = N/4, center = True: the sliding-window / aligned-stream form of the n_fft-1024 kernel (round 3).
// A wave walks a run of consecutive frames of one clip.  Register j of a lane holds x[s + 4 (lane + 64 j) .. + 3], so the
// next frame is "registers j + 2 of the same lane": the raw samples stay in registers, shifted by two slots per frame,
// and only the 512 new samples (two 16-byte loads per lane) are fetched -- 2 KB of loads per frame instead of 8 KB.
// The spectrum leaves as ONE byte stream in 512-byte aligned blocks (stft1024.hip, template flag AL): rows are 8200
// bytes, row f starts 8 f bytes past a 128-byte line; the output columns of both 512-point FFTs are rotated over the
// lanes by rot = (f 1025) mod 64 = (rot + 1) mod 64 per frame, the radix-2 stage is lane-local and does not care, the
// merge's twiddles (W1024^k / 2, W2048^k) and mirror lane follow the column, and block 16 of a frame (the tail of
// register 15 and the Nyquist bin) is carried into the next frame's block 0.  Sixteen full-line non-temporal stores per
// frame (a seventeenth every 64 frames).
// ---------------------------------------------------------------------------
struct P2kRun {
  const float* x;
  const float* window;
  const float2* tw;
  const float2* tw2k;
  float2* X;
  long long B, L, clip_stride, T, runs_per_clip, frames_per_run;
};

__device__ __forceinline__ void mirror1024_rot(const v2f (&v)[16], v2f (&p)[16], int lane, int rot, int col) {
  const int src = (2 * rot - lane) & 63;
  v2f q[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const v2f a = q[15 - m];
    const v2f b = q[(16 - m) & 15];
    p[m] = (col == 0) ? b : a;
  }
}

// 256 samples (one register slot of the wave) starting at original index i0 of the clip, reflect-padded
__device__ __forceinline__ float4 load_slot2k(const float* clip, long long L, long long i0, int lane) {
  const long long i = i0 + 4 * lane;
  if (i0 >= 0 && i0 + 256 <= L) return *reinterpret_cast<const float4*>(clip + i);     // clip base 16-byte aligned (launcher)
  return make_float4(clip[reflect2k(i, L)], clip[reflect2k(i + 1, L)], clip[reflect2k(i + 2, L)], clip[reflect2k(i + 3, L)]);
}

constexpr int W2KR = 8;     // waves per block of the run kernel: two blocks per CU, four waves per SIMD (126 VGPRs)
__global__ __launch_bounds__(64 * W2KR, 4) void stft2048_run_fwd_kernel(P2kRun p) {
  __shared__ float2 lds_all[W2KR * kFftLdsFloat2PerWave + kTwiddleCount + 1024 + 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W2KR * kFftLdsFloat2PerWave;
  float2* w2tab = tab + kTwiddleCount;
  float4* wintab = reinterpret_cast<float4*>(w2tab + 1024);          // analysis window, 512 float4
  for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * W2KR) tab[i] = twiddle_for_lds<false>(p.tw, i);
  for (int i = threadIdx.x; i < 1024; i += 64 * W2KR) w2tab[i] = p.tw2k[i];
  for (int i = threadIdx.x; i < 512; i += 64 * W2KR) wintab[i] = reinterpret_cast<const float4*>(p.window)[i];
  __syncthreads();

  const long long run = (long long)blockIdx.x * W2KR + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long t0 = r * p.frames_per_run;
  long long t1 = t0 + p.frames_per_run;
  if (t1 > p.T) t1 = p.T;
  if (t0 >= t1) return;
  const float* clip = p.x + b * p.clip_stride;
  const long long L = p.L;
  const LdsTwiddles<false> tw = {tab, lane};
  const v2f hh = {0.5f, 0.5f};

  float4 raw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) raw[j] = load_slot2k(clip, L, t0 * 512 - 1024 + 256 * j, lane);
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(raw[j].x), "+v"(raw[j].y), "+v"(raw[j].z), "+v"(raw[j].w));

  const long long e0 = (b * p.T + t0) * F2K;
  int rot = (int)(e0 & 63);
  float2* sp = p.X + (e0 - rot) + lane;
  v2f carry = {0.f, 0.f};
  bool head = true;
  auto put = [&](float2* dst, v2f val) { __builtin_nontemporal_store(val, reinterpret_cast<v2f*>(dst)); };

  auto frame_body = [&](const float4 (&fresh)[2]) {
    wave_priority<3>();        // transform > stores, as in stft1024.hip
    v2f ze[8], zo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 w = wintab[lane + 64 * j];
      ze[j] = (v2f){raw[j].x * w.x, raw[j].y * w.y};
      zo[j] = (v2f){raw[j].z * w.z, raw[j].w * w.w};
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) raw[j] = raw[j + 2];
    raw[6] = fresh[0];
    raw[7] = fresh[1];
    const int col = (lane - rot) & 63;
    fft512<false>(ze, tw, lds, lane, col);
    fft512<false>(zo, tw, lds, lane, col);
    const LdsTwiddles<false> twc = {tab, col};
    v2f z[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v2f t = cmul_v(zo[m], twc.getr(m));
      const v2f e = ze[m] * hh;
      z[m] = e + t;
      z[m + 8] = e - t;
    }
    v2f pm[16];
    mirror1024_rot(z, pm, lane, rot, col);
    const v2f nyq = {2.0f * (z[0].x - z[0].y), 0.0f};          // X[1024], meaningful on the lane whose column is 0
    const v2f* w2 = reinterpret_cast<const v2f*>(w2tab) + col;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const v2f e = add_conj(z[m], pm[m]);
      const v2f d = sub_conj(z[m], pm[m]);
      z[m] = add_mi(e, cmul_v(d, lds_read_single(w2 + 64 * m)));
    }
    wave_priority<1>();
    const bool lo = lane < rot;
    const v2f s0 = lo ? carry : z[0];
    if (head) {
      if (!lo) put(sp, s0);
      head = false;
    } else {
      put(sp, s0);
    }
#pragma unroll
    for (int j = 1; j < 16; ++j) put(sp + 64 * j, lo ? z[j - 1] : z[j]);
    carry = lo ? z[15] : nyq;
    if (rot == 63) {
      put(sp + 1024, carry);
      sp += 1088;
      rot = 0;
    } else {
      sp += 1024;
      ++rot;
    }
    wave_priority<0>();
  };

  long long t = t0;
  long long t_fast_end = (L >= 1536) ? (L - 1536) / 512 + 1 : 0;     // first t whose successor's new samples need reflection
  if (t_fast_end > t1 - 1) t_fast_end = t1 - 1;
  if (t < t_fast_end) {
    const float4* nsrc = reinterpret_cast<const float4*>(clip + t * 512 + 1024) + lane;
    for (; t < t_fast_end; ++t) {
      float4 fresh[2];
      fresh[0] = nsrc[0];
      fresh[1] = nsrc[64];
      nsrc += 128;
      frame_body(fresh);
    }
  }
  for (; t < t1; ++t) {
    float4 fresh[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (t + 1 < t1) {
      fresh[0] = load_slot2k(clip, L, t * 512 + 1024, lane);
      fresh[1] = load_slot2k(clip, L, t * 512 + 1280, lane);
    }
    frame_body(fresh);
  }
  if (lane < rot) put(sp, carry);
}

__device__ __forceinline__ void sincos_big2k(float phase, float& s, float& c) {
  double t = (double)phase * 0.15915494309189533577;  // 1 / (2 pi)
  t -= rint(t);
  const float r = (float)t;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

// irfft(X) * window, frames out (the overlap-add is stft_generic.hip's gather): complex or polar input
template <bool POLAR>
__global__ __launch_bounds__(64 * W2K, 3) void irfft2048_frames_kernel(P2k p) {
  __shared__ float2 lds_all[W2K * kFftLdsFloat2PerWave + kTwiddleCount + 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W2K * kFftLdsFloat2PerWave;
  const v2f* w2 = reinterpret_cast<const v2f*>(tab + kTwiddleCount) + lane;
  stage_tables2k<true>(p, tab, tab + kTwiddleCount);
  const LdsTwiddles<true> tw = {tab, lane};      // tw.getr(m) = W1024^(lane + 64 m)
  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;
  const float4* win4 = reinterpret_cast<const float4*>(p.window);
  const float scale = 1.0f / 2048.0f;
  for (long long f = f_begin + wave; f < f_end; f += W2K) {
    v2f v[16];
    float nyq_re;
    if (POLAR) {
      const float* mrow = p.mag + f * F2K;
      const float* prow = p.phase + f * F2K;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        float sn, cs;
        const float a = mrow[lane + 64 * m];
        sincos_big2k(prow[lane + 64 * m], sn, cs);
        v[m] = (v2f){a * cs, a * sn};
      }
      float sn, cs;
      sincos_big2k(prow[1024], sn, cs);
      nyq_re = mrow[1024] * cs;
    } else {
      const float2* row = p.X + f * F2K;
#pragma unroll
      for (int m = 0; m < 16; ++m) v[m] = to_v(row[lane + 64 * m]);
      nyq_re = row[1024].x;
    }
    if (lane == 0) v[0].y = 0.0f;                       // c2r ignores the imaginary parts of DC and Nyquist
    v2f pm[16];
    mirror1024(v, pm, lane);
    if (lane == 0) pm[0] = (v2f){nyq_re, 0.0f};         // partner of k = 0 is X[1024]
    // Z = E + i O,  E = X + conj X',  O = (X - conj X') conj(W2048^k)   (twice the true value: folded into `scale`)
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const v2f e = add_conj(v[m], pm[m]);
      const v2f d = cmul_conj_v(sub_conj(v[m], pm[m]), w2[64 * m]);
      v[m] = add_pi(e, d);
    }
    // radix-2 backwards: even samples from Z[k] + Z[k+512], odd ones from (Z[k] - Z[k+512]) conj(W1024^k)
    v2f ze[8], zo[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      ze[m] = v[m] + v[m + 8];
      zo[m] = cmul_conj_v(v[m] - v[m + 8], tw.getr(m));
    }
    fft512<true>(ze, tw, lds, lane);
    fft512<true>(zo, tw, lds, lane);
    float4* dst = reinterpret_cast<float4*>(p.y + f * N2K);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 w = win4[lane + 64 * j];
      dst[lane + 64 * j] = make_float4(ze[j].x * (w.x * scale), ze[j].y * (w.y * scale), zo[j].x * (w.z * scale),
                                       zo[j].y * (w.w * scale));
    }
  }
}

// ---------------------------------------------------------------------------
// torch.istft for n_fft = 2048, hop = 256 / 512 / 1024 in one kernel: irfft + window + overlap-add + envelope
// (reference stft.py:120-128, dgt.py:86-93; polar input: stft.py:157-161, dgt.py:152-154).
// A wave walks consecutive frames of one clip.  A frame is 8 register slots of 256 samples (float4 per lane); a hop
// is HS = hop / 256 slots, so the R = 8 / HS frames that overlap a hop are summed in registers: after frame t has
// been added, block t of the padded signal (samples [t hop, (t + 1) hop)) is complete -- it is divided by the window
// envelope of the frames that exist around it (the 2^R x hop table of at_istft_envelope_table) and stored once.
// Output sample s is padded sample s + 1024 (center = True trims n_fft / 2 at both ends): block c is output hop
// c - 1024 / hop.  A run of blocks [c0, c1) starts R - 1 frames early (warm-up) to have its first block complete.
// ---------------------------------------------------------------------------
struct P2kOla {
  const float2* X;       // (B*T, 1025) complex64, or null
  const float* mag;      // polar input
  const float* phase;
  const float* window;   // 2048 synthesis window samples
  const float* env;      // 2^R x hop
  const float2* tw;
  const float2* tw2k;
  float* y;              // (B, hop (T - 1))
  long long B, T, runs_per_clip, blocks_per_run;
};

template <bool POLAR, int HS>
__global__ __launch_bounds__(64 * W2K, 2) void istft2048_ola_kernel(P2kOla p) {
  constexpr int HOP = 256 * HS, R = 8 / HS, LEAD = 1024 / HOP;      // LEAD: blocks trimmed at the front
  __shared__ float2 lds_all[W2K * kFftLdsFloat2PerWave + kTwiddleCount + 1024 + 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W2K * kFftLdsFloat2PerWave;
  const v2f* w2 = reinterpret_cast<const v2f*>(tab + kTwiddleCount) + lane;
  for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * W2K) tab[i] = twiddle_for_lds<true>(p.tw, i);
  for (int i = threadIdx.x; i < 1024; i += 64 * W2K) tab[kTwiddleCount + i] = p.tw2k[i];
  // the synthesis window with the transform's 1/2048 folded in, shared by the block's waves: read from global memory
  // per frame it was as many bytes through the vector-memory path as the spectrum row itself
  float4* wintab = reinterpret_cast<float4*>(tab + kTwiddleCount + 1024);
  for (int i = threadIdx.x; i < 512; i += 64 * W2K) {
    const float4 w = reinterpret_cast<const float4*>(p.window)[i];
    const float sc = 1.0f / 2048.0f;
    wintab[i] = make_float4(w.x * sc, w.y * sc, w.z * sc, w.w * sc);
  }
  __syncthreads();
  const LdsTwiddles<true> tw = {tab, lane};
  const long long run = (long long)blockIdx.x * W2K + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long T = p.T;
  // output hops q = 0 .. T - 2 are blocks c = q + LEAD
  const long long c0 = LEAD + r * p.blocks_per_run;
  long long c1 = c0 + p.blocks_per_run;
  if (c1 > LEAD + T - 1) c1 = LEAD + T - 1;
  if (c0 >= c1) return;
  const float4* env4 = reinterpret_cast<const float4*>(p.env);
  float4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float* yclip = p.y + b * (HOP * (T - 1));

  // Complex input: the next frame's row is requested before the current one is transformed (one frame = 8 KB per wave
  // in flight; the index is clamped into the clip so that the load is unconditional -- a conditional load would make
  // the compiler drain vmcnt on the spot).  Until round 3 every frame's loads were issued and consumed back to back.
  v2f nxt[16];
  float nxt_nyq = 0.f;
  auto request = [&](long long t) {
    const long long tc = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    const float2* row = p.X + (b * T + tc) * F2K;
#pragma unroll
    for (int m = 0; m < 16; ++m) nxt[m] = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(row) + lane + 64 * m);
    nxt_nyq = reinterpret_cast<const float*>(row + 1024)[0];
  };
  if constexpr (!POLAR) request(c0 - (R - 1));
  // reciprocal of the fully overlapped envelope: one division per wave instead of four per hop (<= 1 ulp from acc / e)
  float4 rcp_full[HS];
#pragma unroll
  for (int j = 0; j < HS; ++j) {
    const float4 e = env4[(size_t)((1 << R) - 1) * (HOP / 4) + lane + 64 * j];
    rcp_full[j] = make_float4(1.0f / e.x, 1.0f / e.y, 1.0f / e.z, 1.0f / e.w);
  }

  for (long long t = c0 - (R - 1); t < c1; ++t) {
    v2f cur[16];
    float cur_nyq = 0.f;
    if constexpr (!POLAR) {
#pragma unroll
      for (int m = 0; m < 16; ++m) cur[m] = nxt[m];
      cur_nyq = nxt_nyq;
      request(t + 1);
    }
    if (t >= 0 && t < T) {
      const long long f = b * T + t;
      v2f v[16];
      float nyq_re;
      if (POLAR) {
        const float* mrow = p.mag + f * F2K;
        const float* prow = p.phase + f * F2K;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          float sn, cs;
          const float a = mrow[lane + 64 * m];
          sincos_big2k(prow[lane + 64 * m], sn, cs);
          v[m] = (v2f){a * cs, a * sn};
        }
        float sn, cs;
        sincos_big2k(prow[1024], sn, cs);
        nyq_re = mrow[1024] * cs;
      } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = cur[m];
        nyq_re = cur_nyq;
        (void)f;
      }
      if (lane == 0) v[0].y = 0.0f;
      v2f pm[16];
      mirror1024(v, pm, lane);
      if (lane == 0) pm[0] = (v2f){nyq_re, 0.0f};
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const v2f e = add_conj(v[m], pm[m]);
        const v2f d = cmul_conj_v(sub_conj(v[m], pm[m]), w2[64 * m]);
        v[m] = add_pi(e, d);
      }
      v2f ze[8], zo[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        ze[m] = v[m] + v[m + 8];
        zo[m] = cmul_conj_v(v[m] - v[m + 8], tw.getr(m));
      }
      fft512<true>(ze, tw, lds, lane);
      fft512<true>(zo, tw, lds, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 w = wintab[lane + 64 * j];
        acc[j].x += ze[j].x * w.x;
        acc[j].y += ze[j].y * w.y;
        acc[j].z += zo[j].x * w.z;
        acc[j].w += zo[j].y * w.w;
      }
    }
    // block t is complete: frames t - R + 1 .. t are all that cover it
    if (t >= c0) {
      int mask = 0;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const long long ft = t - (R - 1) + q;          // bit q: oldest frame first (at_istft_envelope_table)
        if (ft >= 0 && ft < T) mask |= 1 << q;
      }
      float* dst = yclip + (t - LEAD) * HOP;
      typedef float v4f __attribute__((ext_vector_type(4)));
      if (mask == (1 << R) - 1) {
        // a hop's value must not depend on how the launch cut the clip into runs: every fully overlapped hop takes
        // the reciprocal form, whichever run emits it
#pragma unroll
        for (int j = 0; j < HS; ++j)
          __builtin_nontemporal_store((v4f){acc[j].x * rcp_full[j].x, acc[j].y * rcp_full[j].y, acc[j].z * rcp_full[j].z,
                                            acc[j].w * rcp_full[j].w},
                                      reinterpret_cast<v4f*>(dst + 4 * (lane + 64 * j)));
      } else {
#pragma unroll
        for (int j = 0; j < HS; ++j) {
          const float4 e = env4[(size_t)mask * (HOP / 4) + lane + 64 * j];
          *reinterpret_cast<float4*>(dst + 4 * (lane + 64 * j)) =
              make_float4(acc[j].x / e.x, acc[j].y / e.y, acc[j].z / e.z, acc[j].w / e.w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8 - HS; ++j) acc[j] = acc[j + HS];
#pragma unroll
    for (int j = 8 - HS; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---------------------------------------------------------------------------
// Features-only forward: audio -> normalise(contrast(|X|^p @ bank)) for a banded bank, n_fft = 2048, any hop
// (MelSpectrogram / MFCC at torchaudio's and librosa's usual 2048 / 512: mel.py:43-44, 68-73 behind stft.py:98-104).
// The forward kernel above with the spectrum kept in registers: |X|^p goes into an LDS row of the wave, every lane walks
// the band of its filter(s) (band_bank.h, the walk of mel_banded.hip) and the features leave row-major, or channel-major
// through the eight-frame register window with sector-aligned flushes.  The 2.9 GB spectrum of 1024 clips is never
// written: STFT 1.06 ms + projection 0.78 ms -> one kernel.
// A wave takes a run of consecutive frames (the window wants the frames of a clip in order).
// ---------------------------------------------------------------------------
struct P2kMel {
  const float* x;
  const float* window;
  const float2* tw;
  const float2* tw2k;
  float* feat;           // (B*T, N) or (B, N, T)
  const float* offset;   // Normalize (device scalars) or null
  const float* scale;
  long long L, clip_stride, T, total_frames, frames_per_wave;
  int win_off_f2;        // where the window table starts in dynamic LDS, in float2 units (16-byte aligned)
  BandBank bank;
  int hop, contrast, power2, channel_major, row_floats, table_floats;
  float eps;
};

__device__ __forceinline__ float contrast2k(float v, int mode, float eps) {
  switch (mode) {
    case C_LOG1P: return logf(1.0f + v);
    case C_LOG: return logf(fmaxf(v, eps));
    case C_LOG10: return log10f(fmaxf(v, eps));
    default: return v;
  }
}

// CMW 1 / 2: channel-major output of a bank with that many passes (register window); 0: anything else.  WINLDS: the
// analysis window staged in LDS (when the bank's tables leave 8 KB of the two-blocks-per-CU budget).
template <int CMW, bool WINLDS>
__global__ __launch_bounds__(64 * W2K, 2) void stft2048_mel_kernel(P2kMel p) {
  extern __shared__ __attribute__((aligned(16))) float2 lds_all[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W2K * kFftLdsFloat2PerWave;
  const v2f* w2 = reinterpret_cast<const v2f*>(tab + kTwiddleCount) + lane;
  float* rows = reinterpret_cast<float*>(tab + kTwiddleCount + 1024);
  float* absrow = rows + wave * p.row_floats;
  float* wlds = rows + W2K * p.row_floats;
  int* lane_tab = reinterpret_cast<int*>(wlds + p.table_floats);
  for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * W2K) tab[i] = twiddle_for_lds<false>(p.tw, i);
  for (int i = threadIdx.x; i < 1024; i += 64 * W2K) tab[kTwiddleCount + i] = p.tw2k[i];
  for (int i = threadIdx.x; i < p.table_floats; i += 64 * W2K) wlds[i] = p.bank.weights[i];
  for (int i = threadIdx.x; i < 64 * p.bank.n_passes; i += 64 * W2K) {
    lane_tab[i] = p.bank.lane_start[i];
    lane_tab[64 * p.bank.n_passes + i] = p.bank.lane_filter[i];
  }
  for (int k = 1024 + lane; k < p.row_floats; k += 64) absrow[k] = 0.0f;     // bin 1024 is rewritten per frame
  // the analysis window, shared by the block's waves (round 3): read from global memory per frame it was 8 KB per frame
  // through the vector-memory path, as much as the frame's samples
  float4* wintab = reinterpret_cast<float4*>(lds_all + (WINLDS ? p.win_off_f2 : 0));
  if constexpr (WINLDS)
    for (int i = threadIdx.x; i < 512; i += 64 * W2K) wintab[i] = reinterpret_cast<const float4*>(p.window)[i];
  const float4* win4 = reinterpret_cast<const float4*>(p.window);
  __syncthreads();
  const LdsTwiddles<false> tw = {tab, lane};
  const long long f_begin = ((long long)blockIdx.x * W2K + wave) * p.frames_per_wave;
  long long f_end = f_begin + p.frames_per_wave;
  if (f_end > p.total_frames) f_end = p.total_frames;
  if (f_begin >= f_end) return;
  const v2f hh = {0.5f, 0.5f};
  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }
  // forward kernel's frame loader on this struct's fields
  P2k lp = {};
  lp.x = p.x; lp.L = p.L; lp.clip_stride = p.clip_stride; lp.T = p.T; lp.hop = p.hop; lp.center = 1;

  float cm[CMW > 0 ? CMW : 1][8];
  long long e_next[CMW > 0 ? CMW : 1];
  int held[CMW > 0 ? CMW : 1];
#pragma unroll
  for (int q = 0; q < (CMW > 0 ? CMW : 1); ++q) {
    held[q] = 0;
    e_next[q] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cm[q][k] = 0.f;
  }
  bool e_valid = false;
  long long cb = f_begin / p.T, ct = f_begin - cb * p.T;

  float4 nxt[8];
  load_frame2k(lp, f_begin, lane, nxt);
  for (long long f = f_begin; f < f_end; ++f) {
    wave_priority<3>();        // transform > epilogue
    v2f ze[8], zo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 w = WINLDS ? wintab[lane + 64 * j] : win4[lane + 64 * j];
      ze[j] = (v2f){nxt[j].x * w.x, nxt[j].y * w.y};
      zo[j] = (v2f){nxt[j].z * w.z, nxt[j].w * w.w};
    }
    if (f + 1 < f_end) load_frame2k(lp, f + 1, lane, nxt);
    fft512<false>(ze, tw, lds, lane);
    fft512<false>(zo, tw, lds, lane);
    v2f z[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v2f t = cmul_v(zo[m], tw.getr(m));
      const v2f e = ze[m] * hh;
      z[m] = e + t;
      z[m + 8] = e - t;
    }
    v2f pm[16];
    mirror1024(z, pm, lane);
    const float nyq = 2.0f * (z[0].x - z[0].y);          // X[1024] (lane 0), real
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const v2f e = add_conj(z[m], pm[m]);
      const v2f d = sub_conj(z[m], pm[m]);
      const v2f xk = add_mi(e, cmul_v(d, w2[64 * m]));
      const float s2 = fmaf(xk.x, xk.x, xk.y * xk.y);
      absrow[lane + 64 * m] = p.power2 ? s2 : __builtin_amdgcn_sqrtf(s2);
    }
    if (lane == 0) absrow[1024] = p.power2 ? nyq * nyq : fabsf(nyq);
    wave_lds_sync();
    wave_priority<0>();

    const long long b = cb, t = ct;
    if (++ct == p.T) {
      ct = 0;
      ++cb;
    }
    const float4* w = reinterpret_cast<const float4*>(wlds) + lane;
    if constexpr (CMW > 0) {
      int fq[CMW];
      const bool last_of_run = (t == p.T - 1) || (f == f_end - 1);
#pragma unroll
      for (int q = 0; q < CMW; ++q) {
        fq[q] = lane_tab[(CMW + q) * 64 + lane];
        const float4* a = reinterpret_cast<const float4*>(absrow + lane_tab[q * 64 + lane]);
        v2f acc2 = {0.f, 0.f};
        const int quads = p.bank.pass_len[q] >> 2;
        int j = 0;
        for (; j + 4 <= quads; j += 4) {          // four steps' reads ahead of the multiply-adds
          float4 av[4], wv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = a[j + u];
            wv[u] = w[(j + u) * 64];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc2 = __builtin_elementwise_fma((v2f){av[u].x, av[u].y}, (v2f){wv[u].x, wv[u].y}, acc2);
            acc2 = __builtin_elementwise_fma((v2f){av[u].z, av[u].w}, (v2f){wv[u].z, wv[u].w}, acc2);
          }
        }
        for (; j < quads; ++j) {
          const float4 av = a[j], wv = w[j * 64];
          acc2 = __builtin_elementwise_fma((v2f){av.x, av.y}, (v2f){wv.x, wv.y}, acc2);
          acc2 = __builtin_elementwise_fma((v2f){av.z, av.w}, (v2f){wv.z, wv.w}, acc2);
        }
        w += quads * 64;
        float acc = contrast2k(acc2.x + acc2.y, p.contrast, p.eps);
        if (p.offset) acc = (acc - off) / sc;
#pragma unroll
        for (int k = 0; k < 7; ++k) cm[q][k] = cm[q][k + 1];
        cm[q][7] = acc;
        ++held[q];
        if (fq[q] >= 0) {
          if (!e_valid) e_next[q] = (b * p.bank.n_filters + fq[q]) * p.T + t + 1;   // one past frame t in this lane's row
          const long long e = e_next[q];
          e_next[q] = e + ((t == p.T - 1) ? (long long)(p.bank.n_filters - 1) * p.T + 1 : 1);
          if ((e & 7) == 0 || last_of_run) {            // whole 32-byte sectors (see mel_banded.hip)
            float* dst = p.feat + e - 8;
            if (held[q] >= 8) {
              if ((e & 3) == 0) {
                reinterpret_cast<float4*>(dst)[0] = make_float4(cm[q][0], cm[q][1], cm[q][2], cm[q][3]);
                reinterpret_cast<float4*>(dst)[1] = make_float4(cm[q][4], cm[q][5], cm[q][6], cm[q][7]);
              } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) dst[k] = cm[q][k];
              }
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k)
                if (k >= 8 - held[q]) dst[k] = cm[q][k];
            }
            held[q] = 0;
          }
        } else if (last_of_run) {
          held[q] = 0;
        }
      }
      e_valid = true;
    } else {
      for (int q = 0; q < p.bank.n_passes; ++q) {
        const int filt = lane_tab[(p.bank.n_passes + q) * 64 + lane];
        const float4* a = reinterpret_cast<const float4*>(absrow + lane_tab[q * 64 + lane]);
        v2f acc2 = {0.f, 0.f};
        const int quads = p.bank.pass_len[q] >> 2;
        int j = 0;
        for (; j + 4 <= quads; j += 4) {          // four steps' reads ahead of the multiply-adds
          float4 av[4], wv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = a[j + u];
            wv[u] = w[(j + u) * 64];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc2 = __builtin_elementwise_fma((v2f){av[u].x, av[u].y}, (v2f){wv[u].x, wv[u].y}, acc2);
            acc2 = __builtin_elementwise_fma((v2f){av[u].z, av[u].w}, (v2f){wv[u].z, wv[u].w}, acc2);
          }
        }
        for (; j < quads; ++j) {
          const float4 av = a[j], wv = w[j * 64];
          acc2 = __builtin_elementwise_fma((v2f){av.x, av.y}, (v2f){wv.x, wv.y}, acc2);
          acc2 = __builtin_elementwise_fma((v2f){av.z, av.w}, (v2f){wv.z, wv.w}, acc2);
        }
        w += quads * 64;
        if (filt >= 0) {
          float acc = contrast2k(acc2.x + acc2.y, p.contrast, p.eps);
          if (p.offset) acc = (acc - off) / sc;
          if (p.channel_major) p.feat[(b * p.bank.n_filters + filt) * p.T + t] = acc;
          else p.feat[f * p.bank.n_filters + filt] = acc;
        }
      }
    }
    wave_lds_sync();
  }
}

int launch_stft2048_mel(const float* x, long long B, long long L, long long clip_stride, long long T, int hop,
                        const float* window, const float2* tw, const float2* tw2k, const BandBank* bank, int contrast,
                        int power2, const float* offset, const float* scale, float eps, float* feat, int channel_major,
                        hipStream_t stream) {
  const long long nframes = B * T;
  if (nframes == 0) return 0;
  P2kMel p = {};
  p.x = x; p.window = window; p.tw = tw; p.tw2k = tw2k; p.feat = feat; p.offset = offset; p.scale = scale;
  p.L = L; p.clip_stride = clip_stride; p.T = T; p.total_frames = nframes; p.bank = *bank; p.hop = hop;
  p.contrast = contrast; p.power2 = power2; p.channel_major = channel_major; p.eps = eps;
  int max_walk = 0, table_floats = 0;
  for (int q = 0; q < bank->n_passes; ++q) {
    max_walk = bank->pass_len[q] > max_walk ? bank->pass_len[q] : max_walk;
    table_floats += 64 * bank->pass_len[q];
  }
  p.table_floats = table_floats;
  p.row_floats = (F2K + max_walk + 63) / 64 * 64;      // a walk that starts on the last bins runs into zeros
  size_t lds = sizeof(float2) * (size_t)(W2K * kFftLdsFloat2PerWave + kTwiddleCount + 1024) +
               sizeof(float) * ((size_t)W2K * p.row_floats + table_floats) + sizeof(int) * (size_t)2 * 64 * bank->n_passes;
  if (lds > 80 * 1024) return -2;                       // two workgroups per CU or not at all
  lds = (lds + 15) / 16 * 16;
  const bool winlds = lds + 8192 <= 80 * 1024;          // the analysis window (512 float4) too, when it fits
  if (winlds) {
    p.win_off_f2 = (int)(lds / sizeof(float2));
    lds += 8192;
  }
  // runs of consecutive frames per wave: long enough for the window and the table staging, short enough to fill the chip
  long long fpw = (nframes + 256LL * 8 * W2K - 1) / (256LL * 8 * W2K);
  if (fpw < 8) fpw = 8;
  p.frames_per_wave = fpw;
  const long long waves = (nframes + fpw - 1) / fpw;
  const unsigned grid = (unsigned)((waves + W2K - 1) / W2K);
  void (*kernel)(P2kMel) = winlds ? stft2048_mel_kernel<0, true> : stft2048_mel_kernel<0, false>;
  if (channel_major && bank->n_passes == 1) kernel = winlds ? stft2048_mel_kernel<1, true> : stft2048_mel_kernel<1, false>;
  else if (channel_major && bank->n_passes == 2) kernel = winlds ? stft2048_mel_kernel<2, true> : stft2048_mel_kernel<2, false>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return -5;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * W2K), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_istft2048_ola(const float2* X, const float* mag, const float* phase, long long B, long long T, int hop,
                         const float* window, const float* env, const float2* tw, const float2* tw2k, float* y,
                         hipStream_t stream) {
  if (B == 0 || T <= 1) return 0;
  P2kOla p = {X, mag, phase, window, env, tw, tw2k, y, B, T, 0, 0};
  const long long blocks = T - 1;                        // output hops per clip
  // runs long enough that the R - 1 warm-up frames stay a small share, short enough to fill the chip
  long long runs = (B >= 2048) ? 1 : (2048 + B - 1) / B;
  long long per = (blocks + runs - 1) / runs;
  const long long min_per = 8;
  if (per < min_per) per = min_per < blocks ? min_per : blocks;
  runs = (blocks + per - 1) / per;
  p.runs_per_clip = runs;
  p.blocks_per_run = per;
  const long long waves = B * runs;
  const unsigned grid = (unsigned)((waves + W2K - 1) / W2K);
#define OLA2K(POLAR_, HS_) hipLaunchKernelGGL((istft2048_ola_kernel<POLAR_, HS_>), dim3(grid), dim3(64 * W2K), 0, stream, p)
  const bool polar = (X == nullptr);
  if (hop == 256) { if (polar) OLA2K(true, 1); else OLA2K(false, 1); }
  else if (hop == 512) { if (polar) OLA2K(true, 2); else OLA2K(false, 2); }
  else if (hop == 1024) { if (polar) OLA2K(true, 4); else OLA2K(false, 4); }
  else return -2;
#undef OLA2K
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

static long long frames_per_block_2k(long long nframes) {
  const long long max_blocks = 256LL * 8;
  long long fpb = (nframes + max_blocks - 1) / max_blocks;
  fpb = ((fpb + W2K - 1) / W2K) * W2K;
  return fpb < W2K ? W2K : fpb;
}

int launch_stft2048_fwd(const float* x, long long B, long long L, long long clip_stride, long long T, int hop, int center,
                        const float* window, const float2* tw, const float2* tw2k, float2* out, float* phase,
                        hipStream_t stream) {
  const long long nframes = B * T;
  if (nframes == 0) return 0;
  P2k p = {};
  p.x = x; p.window = window; p.tw = tw; p.tw2k = tw2k; p.X = out; p.phase_out = phase;
  p.L = L; p.clip_stride = clip_stride; p.T = T; p.total_frames = nframes; p.hop = hop; p.center = center;
  // the sliding-window / aligned-stream kernel: torch.stft's framing at hop n/4, 16-byte aligned clips, a 512-byte
  // aligned output (torch allocations are), no phase side output
  if (center && hop == 512 && !phase && L >= 2048 && (clip_stride & 3) == 0 && (((uintptr_t)x) & 15) == 0 &&
      (((uintptr_t)out) & 511) == 0 && (((uintptr_t)window) & 15) == 0 && variant(kVarFrameKernels) == 0) {
    P2kRun q = {};
    q.x = x; q.window = window; q.tw = tw; q.tw2k = tw2k; q.X = out;
    q.B = B; q.L = L; q.clip_stride = clip_stride; q.T = T;
    const long long slots = resident_waves(stft2048_run_fwd_kernel, 64 * W2KR, 0);
    q.frames_per_run = plan_units_per_run(B, T, slots, 8, 1);
    q.runs_per_clip = (T + q.frames_per_run - 1) / q.frames_per_run;
    const long long waves = B * q.runs_per_clip;
    hipLaunchKernelGGL(stft2048_run_fwd_kernel, dim3((unsigned)((waves + W2KR - 1) / W2KR)), dim3(64 * W2KR), 0, stream, q);
    return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  p.frames_per_block = frames_per_block_2k(nframes);
  const long long blocks = (nframes + p.frames_per_block - 1) / p.frames_per_block;
  if (phase) hipLaunchKernelGGL(stft2048_fwd_kernel<true>, dim3((unsigned)blocks), dim3(64 * W2K), 0, stream, p);
  else hipLaunchKernelGGL(stft2048_fwd_kernel<false>, dim3((unsigned)blocks), dim3(64 * W2K), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft2048_frames(const float2* X, const float* mag, const float* phase, long long nframes, const float* window,
                            const float2* tw, const float2* tw2k, float* frames, hipStream_t stream) {
  if (nframes == 0) return 0;
  P2k p = {};
  p.X = const_cast<float2*>(X); p.mag = mag; p.phase = phase; p.window = window; p.tw = tw; p.tw2k = tw2k; p.y = frames;
  p.total_frames = nframes;
  p.frames_per_block = frames_per_block_2k(nframes);
  const long long blocks = (nframes + p.frames_per_block - 1) / p.frames_per_block;
  if (X) hipLaunchKernelGGL(irfft2048_frames_kernel<false>, dim3((unsigned)blocks), dim3(64 * W2K), 0, stream, p);
  else hipLaunchKernelGGL(irfft2048_frames_kernel<true>, dim3((unsigned)blocks), dim3(64 * W2K), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
