// stft4096.hip -- n_fft = 4096 on the one-wavefront register FFT core (fft512.h).
//
// Replaces, for n_fft = 4096 (any hop):
//   torch.stft(...).transpose(-2,-1)          reference transforms/stft.py:98-104, dgt.py:64-70
//   torch.fft.rfft(x*window) on frames        stft.py:249-253, dgt.py:285-289
//   torch.fft.irfft(X) * inv_window           stft.py:260-266, dgt.py:296-302; the frames of torch.istft
//                                             (stft.py:120-128), overlap-added by stft_generic.hip's gather
// Until this file the size ran on the workgroup-per-frame LDS Stockham kernel of stft_generic.hip (~1.3 TB/s).
//
// A 4096-point real transform is a 2048-point complex FFT of z[n] = x[2n] + i x[2n+1] plus the real split; the
// 2048-point FFT is four 512-point FFTs (the wave-level radix-8 core: 8 complex points per lane) of the samples
// z[4m + r], r = 0..3, plus one radix-4 stage:
//   z_r[m] = z[4m + r] = x[8m + 2r] + i x[8m + 2r + 1]     -> two float4 loads per lane and register hold
//                                                              (z0, z1) and (z2, z3) of the same m
//   Z[k + 512 q] = sum_r (-i)^(r q) W2048^(r k) Z_r[k],  k = lane + 64 m  (lane-local: registers m + 8 q)
//   X[k] = (Z[k] + conj Z[2048-k])/2 - (i/2) W4096^k (Z[k] - conj Z[2048-k]),  k = 0 .. 2048
// The mirror partner Z[2048-k] lives in lane 64-lane, register 31-m (lane 0: its own register 32-m).  The inverse
// runs the same steps backwards.  One wave = one frame at a time, frames of a block interleaved over its four waves.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"
#include "fastmath.h"
#include "fft512.h"
#include "run_plan.h"
#include "variants.h"
#include <stdlib.h>

namespace at_hip {

constexpr int N4K = 4096;
constexpr int F4K = N4K / 2 + 1;     // 2049
constexpr int W4K = 4;               // waves per block
constexpr int kTab4k = 2048 + 3 * 512;   // W4096^k, k < 2048; then W2048^(r k), r = 1..3, k < 512 (capi.hip)

struct P4k {
  const float* x;        // forward: audio, clip b at x + b*clip_stride
  const float* window;   // 4096 samples (analysis or synthesis)
  const float2* tw;      // fft512 twiddle table (capi.hip)
  const float2* tw4k;    // kTab4k entries
  float2* X;             // (frames, 2049) complex64: forward output / inverse input
  const float* mag;      // inverse, polar input
  const float* phase;
  float* phase_out;      // forward: optional angle output (frames, 2049)
  float* y;              // inverse: (frames, 4096) windowed time frames
  long long L, clip_stride, T, total_frames, frames_per_block;
  int hop, center;
};

__device__ __forceinline__ long long reflect4k(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// (qa[j], qb[j]) = x[s + 8 (lane + 64 j) .. + 7] of frame f (reflect padding with center, zero padding without)
__device__ __forceinline__ void load_frame4k(const P4k& p, long long f, int lane, float4 (&qa)[8], float4 (&qb)[8]) {
  const long long b = f / p.T, t = f - b * p.T;
  const float* clip = p.x + b * p.clip_stride;
  const long long start = t * (long long)p.hop - (p.center ? N4K / 2 : 0);
  const bool interior = (start >= 0) && (start + N4K <= p.L);
  if (interior && ((((uintptr_t)(clip + start)) & 15) == 0)) {
    const float4* src = reinterpret_cast<const float4*>(clip + start);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qa[j] = src[2 * (lane + 64 * j)];
      qb[j] = src[2 * (lane + 64 * j) + 1];
    }
  } else if (interior) {
    const float* src = clip + start;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 8 * (lane + 64 * j);
      qa[j] = make_float4(src[i], src[i + 1], src[i + 2], src[i + 3]);
      qb[j] = make_float4(src[i + 4], src[i + 5], src[i + 6], src[i + 7]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long i0 = start + 8 * (lane + 64 * j);
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const long long i = i0 + c;
        if (p.center) v[c] = clip[reflect4k(i, p.L)];
        else v[c] = (i >= 0 && i < p.L) ? clip[i] : 0.0f;     // zero padding past the end (utils/misc.py:156)
      }
      qa[j] = make_float4(v[0], v[1], v[2], v[3]);
      qb[j] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// mirror partners P[m] = Z[(2048 - (lane + 64 m)) mod 2048] of the 32 registers
__device__ __forceinline__ void mirror2048(const v2f (&v)[32], v2f (&p)[32], int lane) {
  const int src = (64 - lane) & 63;
  v2f q[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const v2f a = q[31 - m];            // lane > 0: lane 64 - lane, register 31 - m
    const v2f b = q[(32 - m) & 31];     // lane 0: own register (32 - m) mod 32
    p[m] = (lane == 0) ? b : a;
  }
}

// Constant tables in LDS, shared by the block's four waves: the fft512 twiddles (11 KB), W4096^k (16 KB; halved in the
// forward kernel: the real split wants W / 2) and the radix-4 twiddles W2048^(r k) (12 KB).
template <bool INV>
__device__ __forceinline__ void stage_tables4k(const float2* tw, const float2* tw4k, float2* tab, float2* t4) {
  for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * W4K) tab[i] = tw[i];
  for (int i = threadIdx.x; i < kTab4k; i += 64 * W4K) {
    float2 a = tw4k[i];
    if (!INV && i < 2048) a = make_float2(0.5f * a.x, 0.5f * a.y);
    t4[i] = a;
  }
  __syncthreads();
}

constexpr int kLds4k = W4K * kFftLdsFloat2PerWave + kTwiddleCount + kTab4k;

template <bool WRITE_PHASE>
__global__ __launch_bounds__(64 * W4K, 2) void stft4096_fwd_kernel(P4k p) {
  __shared__ float2 lds_all[kLds4k];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W4K * kFftLdsFloat2PerWave;
  float2* t4 = tab + kTwiddleCount;
  const v2f* wh = reinterpret_cast<const v2f*>(t4) + lane;            // wh[64 m] = W4096^(lane + 64 m) / 2
  const v2f* wr = reinterpret_cast<const v2f*>(t4 + 2048) + lane;     // wr[(r - 1) * 512 + 64 m] = W2048^(r (lane + 64 m))
  stage_tables4k<false>(p.tw, p.tw4k, tab, t4);
  const LdsTwiddles<false> tw = {tab, lane};
  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;
  const float4* win4 = reinterpret_cast<const float4*>(p.window);
  const v2f hh = {0.5f, 0.5f};

  long long f = f_begin + wave;
  float4 na[8], nb[8];
  if (f < f_end) load_frame4k(p, f, lane, na, nb);
  for (; f < f_end; f += W4K) {
    v2f z0[8], z1[8], z2[8], z3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 wa = win4[2 * (lane + 64 * j)];   // L2-resident (16 KB), re-read per frame
      const float4 wb = win4[2 * (lane + 64 * j) + 1];
      z0[j] = (v2f){na[j].x * wa.x, na[j].y * wa.y};
      z1[j] = (v2f){na[j].z * wa.z, na[j].w * wa.w};
      z2[j] = (v2f){nb[j].x * wb.x, nb[j].y * wb.y};
      z3[j] = (v2f){nb[j].z * wb.z, nb[j].w * wb.w};
    }
    if (f + W4K < f_end) load_frame4k(p, f + W4K, lane, na, nb);
    fft512<false>(z0, tw, lds, lane);
    fft512<false>(z1, tw, lds, lane);
    fft512<false>(z2, tw, lds, lane);
    fft512<false>(z3, tw, lds, lane);
    // radix-4: registers m, m + 8, m + 16, m + 24 hold Z[k], Z[k + 512], Z[k + 1024], Z[k + 1536]
    v2f z[32];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v2f t1 = cmul_v(z1[m], wr[64 * m]);
      const v2f t2 = cmul_v(z2[m], wr[512 + 64 * m]);
      const v2f t3 = cmul_v(z3[m], wr[1024 + 64 * m]);
      const v2f a = z0[m] + t2, b = z0[m] - t2, c = t1 + t3, d = t1 - t3;
      z[m] = a + c;
      z[m + 16] = a - c;
      z[m + 8] = add_mi(b, d);        // b - i d
      z[m + 24] = add_pi(b, d);       // b + i d
    }
    v2f pm[32];
    mirror2048(z, pm, lane);
    // X[k] = (Z[k] + conj Z')/2 - i (W4096^k / 2) (Z[k] - conj Z'),  Z' = Z[2048 - k]  (k = 0: Z' = Z[0], X[0] real)
    const float2 nyq = make_float2(z[0].x - z[0].y, 0.0f);      // X[2048] = Re Z[0] - Im Z[0] (lane 0)
    float2* row = p.X + f * F4K;
    float* prow = WRITE_PHASE ? p.phase_out + f * F4K : nullptr;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const v2f e = add_conj(z[m], pm[m]);
      const v2f d = sub_conj(z[m], pm[m]);
      const v2f xk = scale_add_mi(e, hh, cmul_v(d, wh[64 * m]));      // e / 2 - i (W / 2) d
      row[lane + 64 * m] = to_f2(xk);
      if (WRITE_PHASE) prow[lane + 64 * m] = fast_atan2f(xk.y, xk.x);
    }
    if (lane == 0) {
      row[2048] = nyq;
      if (WRITE_PHASE) prow[2048] = fast_atan2f(nyq.y, nyq.x);
    }
  }
}


// ---------------------------------------------------------------------------
// forward, hop = 1024 = N/4, center = True: sliding window in registers + aligned stream stores (round 3; the scheme of
// stft1024.hip's AL kernel and stft2048.hip's run kernel).  Register slot j of a lane holds x[s + 8 (lane + 64 j) .. + 7]
// (two float4): the next frame is "slots j + 2 of the same lane", so the raw samples stay in registers and only the
// 1024 new samples (four 16-byte loads per lane) are fetched per frame -- 4 KB instead of 16 KB.  Rows are 16 392 bytes
// (8 f bytes past a 128-byte line): the output columns of the four 512-point FFTs are rotated over the lanes by rot =
// (f 2049) mod 64, the radix-4 stage is lane-local, the twiddles of the radix-4 stage and of the merge and the mirror
// lane follow the column, block 32 of a frame (the tail of register 31, then the Nyquist bin) is carried into the next
// frame's block 0: 32 full-line non-temporal stores per frame.
// ---------------------------------------------------------------------------
struct P4kRun {
  const float* x;
  const float* window;
  const float2* tw;
  const float2* tw4k;
  float2* X;
  long long B, L, clip_stride, T, runs_per_clip, frames_per_run;
};

__device__ __forceinline__ void mirror2048_rot(const v2f (&v)[32], v2f (&p)[32], int lane, int rot, int col) {
  const int src = (2 * rot - lane) & 63;
  v2f q[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const v2f a = q[31 - m];
    const v2f b = q[(32 - m) & 31];
    p[m] = (col == 0) ? b : a;
  }
}

// 512 samples (one register slot of the wave) starting at original index i0 of the clip, reflect-padded
__device__ __forceinline__ void load_slot4k(const float* clip, long long L, long long i0, int lane, float4& a, float4& b) {
  const long long i = i0 + 8 * lane;
  if (i0 >= 0 && i0 + 512 <= L) {                                 // clip base 16-byte aligned (launcher)
    a = *reinterpret_cast<const float4*>(clip + i);
    b = *reinterpret_cast<const float4*>(clip + i + 4);
    return;
  }
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = clip[reflect4k(i + c, L)];
  a = make_float4(v[0], v[1], v[2], v[3]);
  b = make_float4(v[4], v[5], v[6], v[7]);
}

constexpr int kLds4kRun = kLds4k + 2048;          // + the analysis window (1024 float4)

__global__ __launch_bounds__(64 * W4K, 2) void stft4096_run_fwd_kernel(P4kRun p) {
  __shared__ float2 lds_all[kLds4kRun];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W4K * kFftLdsFloat2PerWave;
  float2* t4 = tab + kTwiddleCount;
  float4* wintab = reinterpret_cast<float4*>(t4 + kTab4k);
  for (int i = threadIdx.x; i < 1024; i += 64 * W4K) wintab[i] = reinterpret_cast<const float4*>(p.window)[i];
  stage_tables4k<false>(p.tw, p.tw4k, tab, t4);       // ends with __syncthreads()

  const long long run = (long long)blockIdx.x * W4K + wave;
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

  float4 ra[8], rb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) load_slot4k(clip, L, t0 * 1024 - 2048 + 512 * j, lane, ra[j], rb[j]);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    asm volatile("" : "+v"(ra[j].x), "+v"(ra[j].y), "+v"(ra[j].z), "+v"(ra[j].w));
    asm volatile("" : "+v"(rb[j].x), "+v"(rb[j].y), "+v"(rb[j].z), "+v"(rb[j].w));
  }

  const long long e0 = (b * p.T + t0) * F4K;
  int rot = (int)(e0 & 63);
  float2* sp = p.X + (e0 - rot) + lane;
  v2f carry = {0.f, 0.f};
  bool head = true;
  auto put = [&](float2* dst, v2f val) { __builtin_nontemporal_store(val, reinterpret_cast<v2f*>(dst)); };

  auto frame_body = [&](const float4 (&fa)[2], const float4 (&fb)[2]) {
    wave_priority<3>();        // transform > stores, as in stft1024.hip
    v2f z0[8], z1[8], z2[8], z3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 wa = wintab[2 * (lane + 64 * j)];
      const float4 wb = wintab[2 * (lane + 64 * j) + 1];
      z0[j] = (v2f){ra[j].x * wa.x, ra[j].y * wa.y};
      z1[j] = (v2f){ra[j].z * wa.z, ra[j].w * wa.w};
      z2[j] = (v2f){rb[j].x * wb.x, rb[j].y * wb.y};
      z3[j] = (v2f){rb[j].z * wb.z, rb[j].w * wb.w};
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      ra[j] = ra[j + 2];
      rb[j] = rb[j + 2];
    }
    ra[6] = fa[0]; rb[6] = fb[0];
    ra[7] = fa[1]; rb[7] = fb[1];
    const int col = (lane - rot) & 63;
    fft512<false>(z0, tw, lds, lane, col);
    fft512<false>(z1, tw, lds, lane, col);
    fft512<false>(z2, tw, lds, lane, col);
    fft512<false>(z3, tw, lds, lane, col);
    const v2f* wh = reinterpret_cast<const v2f*>(t4) + col;            // wh[64 m] = W4096^(col + 64 m) / 2
    const v2f* wr = reinterpret_cast<const v2f*>(t4 + 2048) + col;     // wr[(r - 1) 512 + 64 m] = W2048^(r (col + 64 m))
    v2f z[32];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v2f t1 = cmul_v(z1[m], wr[64 * m]);
      const v2f t2 = cmul_v(z2[m], wr[512 + 64 * m]);
      const v2f t3 = cmul_v(z3[m], wr[1024 + 64 * m]);
      const v2f a = z0[m] + t2, bb = z0[m] - t2, c = t1 + t3, d = t1 - t3;
      z[m] = a + c;
      z[m + 16] = a - c;
      z[m + 8] = add_mi(bb, d);
      z[m + 24] = add_pi(bb, d);
    }
    v2f pm[32];
    mirror2048_rot(z, pm, lane, rot, col);
    const v2f nyq = {z[0].x - z[0].y, 0.0f};                           // X[2048], on the lane whose column is 0
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const v2f e = add_conj(z[m], pm[m]);
      const v2f d = sub_conj(z[m], pm[m]);
      z[m] = scale_add_mi(e, hh, cmul_v(d, wh[64 * m]));
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
    for (int j = 1; j < 32; ++j) put(sp + 64 * j, lo ? z[j - 1] : z[j]);
    carry = lo ? z[31] : nyq;
    if (rot == 63) {
      put(sp + 2048, carry);
      sp += 2112;
      rot = 0;
    } else {
      sp += 2048;
      ++rot;
    }
    wave_priority<0>();
  };

  long long t = t0;
  long long t_fast_end = (L >= 3072) ? (L - 3072) / 1024 + 1 : 0;     // first t whose successor's new samples need reflection
  if (t_fast_end > t1 - 1) t_fast_end = t1 - 1;
  if (t < t_fast_end) {
    const float4* nsrc = reinterpret_cast<const float4*>(clip + t * 1024 + 2048) + 2 * lane;
    for (; t < t_fast_end; ++t) {
      float4 fa[2], fb[2];
      fa[0] = nsrc[0];   fb[0] = nsrc[1];
      fa[1] = nsrc[128]; fb[1] = nsrc[129];
      nsrc += 256;
      frame_body(fa, fb);
    }
  }
  for (; t < t1; ++t) {
    float4 fa[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    float4 fb[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (t + 1 < t1) {
      load_slot4k(clip, L, t * 1024 + 2048, lane, fa[0], fb[0]);
      load_slot4k(clip, L, t * 1024 + 2560, lane, fa[1], fb[1]);
    }
    frame_body(fa, fb);
  }
  if (lane < rot) put(sp, carry);
}

__device__ __forceinline__ void sincos_big4k(float phase, float& s, float& c) {
  double t = (double)phase * 0.15915494309189533577;  // 1 / (2 pi)
  t -= rint(t);
  const float r = (float)t;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

// one-sided spectrum of frame f -> the four sub-transform inputs, ready for fft512<true> (shared by the frames kernel)
template <bool POLAR>
__device__ __forceinline__ void spectrum_to_subffts4k(const float2* X, const float* mag, const float* phase, long long f,
                                                      int lane, const v2f* w4, const v2f* wr, v2f (&y0)[8], v2f (&y1)[8],
                                                      v2f (&y2)[8], v2f (&y3)[8]) {
  v2f v[32];
  float nyq_re;
  if (POLAR) {
    const float* mrow = mag + f * F4K;
    const float* prow = phase + f * F4K;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      float sn, cs;
      const float a = mrow[lane + 64 * m];
      sincos_big4k(prow[lane + 64 * m], sn, cs);
      v[m] = (v2f){a * cs, a * sn};
    }
    float sn, cs;
    sincos_big4k(prow[2048], sn, cs);
    nyq_re = mrow[2048] * cs;
  } else {
    const float2* row = X + f * F4K;
#pragma unroll
    for (int m = 0; m < 32; ++m) v[m] = to_v(row[lane + 64 * m]);
    nyq_re = row[2048].x;
  }
  if (lane == 0) v[0].y = 0.0f;                       // c2r ignores the imaginary parts of DC and Nyquist
  v2f pm[32];
  mirror2048(v, pm, lane);
  if (lane == 0) pm[0] = (v2f){nyq_re, 0.0f};         // partner of k = 0 is X[2048]
  // Z = E + i O,  E = X + conj X',  O = (X - conj X') conj(W4096^k)   (twice the true value: folded into the scale)
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const v2f e = add_conj(v[m], pm[m]);
    const v2f d = cmul_conj_v(sub_conj(v[m], pm[m]), w4[64 * m]);
    v[m] = add_pi(e, d);
  }
  // radix-4 backwards: y_r[k] = conj(W2048^(r k)) sum_q (+i)^(r q) Z[k + 512 q]
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const v2f a = v[m] + v[m + 16], b = v[m] - v[m + 16], c = v[m + 8] + v[m + 24], d = v[m + 8] - v[m + 24];
    y0[m] = a + c;
    y2[m] = cmul_conj_v(a - c, wr[512 + 64 * m]);
    y1[m] = cmul_conj_v(add_pi(b, d), wr[64 * m]);          // b + i d
    y3[m] = cmul_conj_v(add_mi(b, d), wr[1024 + 64 * m]);   // b - i d
  }
}

// irfft(X) * window, frames out (the overlap-add is stft_generic.hip's gather): complex or polar input
template <bool POLAR>
__global__ __launch_bounds__(64 * W4K, 2) void irfft4096_frames_kernel(P4k p) {
  __shared__ float2 lds_all[kLds4k];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W4K * kFftLdsFloat2PerWave;
  float2* t4 = tab + kTwiddleCount;
  const v2f* w4 = reinterpret_cast<const v2f*>(t4) + lane;
  const v2f* wr = reinterpret_cast<const v2f*>(t4 + 2048) + lane;
  stage_tables4k<true>(p.tw, p.tw4k, tab, t4);
  const LdsTwiddles<true> tw = {tab, lane};
  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;
  const float4* win4 = reinterpret_cast<const float4*>(p.window);
  const float scale = 1.0f / 4096.0f;
  for (long long f = f_begin + wave; f < f_end; f += W4K) {
    v2f y0[8], y1[8], y2[8], y3[8];
    spectrum_to_subffts4k<POLAR>(p.X, p.mag, p.phase, f, lane, w4, wr, y0, y1, y2, y3);
    fft512<true>(y0, tw, lds, lane);
    fft512<true>(y1, tw, lds, lane);
    fft512<true>(y2, tw, lds, lane);
    fft512<true>(y3, tw, lds, lane);
    float4* dst = reinterpret_cast<float4*>(p.y + f * N4K);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 wa = win4[2 * (lane + 64 * j)];
      const float4 wb = win4[2 * (lane + 64 * j) + 1];
      dst[2 * (lane + 64 * j)] = make_float4(y0[j].x * (wa.x * scale), y0[j].y * (wa.y * scale), y1[j].x * (wa.z * scale),
                                             y1[j].y * (wa.w * scale));
      dst[2 * (lane + 64 * j) + 1] = make_float4(y2[j].x * (wb.x * scale), y2[j].y * (wb.y * scale),
                                                 y3[j].x * (wb.z * scale), y3[j].y * (wb.w * scale));
    }
  }
}

// ---------------------------------------------------------------------------
// torch.istft for n_fft = 4096, hop = 512 / 1024 / 2048 in one kernel: irfft + window + overlap-add + envelope
// (reference stft.py:120-128, dgt.py:86-93; polar input: stft.py:157-161, dgt.py:152-154).  The scheme of
// istft2048_ola_kernel with 512-sample register slots (two float4 per lane): a hop is HS = hop / 512 slots, the
// R = 8 / HS frames that overlap a hop are summed in registers, a completed block is divided by the window envelope of
// the frames that exist around it (at_istft_envelope_table) and stored once.  Output sample s is padded sample
// s + 2048: block c is output hop c - 2048 / hop.
// ---------------------------------------------------------------------------
struct P4kOla {
  const float2* X;       // (B*T, 2049) complex64, or null
  const float* mag;      // polar input
  const float* phase;
  const float* window;   // 4096 synthesis window samples
  const float* env;      // 2^R x hop
  const float2* tw;
  const float2* tw4k;
  float* y;              // (B, hop (T - 1))
  long long B, T, runs_per_clip, blocks_per_run;
};

template <bool POLAR, int HS>
__global__ __launch_bounds__(64 * W4K, 2) void istft4096_ola_kernel(P4kOla p) {
  constexpr int HOP = 512 * HS, R = 8 / HS, LEAD = 2048 / HOP;      // LEAD: blocks trimmed at the front
  __shared__ float2 lds_all[kLds4k + 2048];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* tab = lds_all + W4K * kFftLdsFloat2PerWave;
  float2* t4 = tab + kTwiddleCount;
  const v2f* w4 = reinterpret_cast<const v2f*>(t4) + lane;
  const v2f* wr = reinterpret_cast<const v2f*>(t4 + 2048) + lane;
  // the synthesis window with the transform's 1/4096 folded in, shared by the block's waves (16 KB): read from global
  // memory per frame it was as many bytes through the vector-memory path as the spectrum row itself
  float4* wintab = reinterpret_cast<float4*>(t4 + kTab4k);
  for (int i = threadIdx.x; i < 1024; i += 64 * W4K) {
    const float4 w = reinterpret_cast<const float4*>(p.window)[i];
    const float sc = 1.0f / 4096.0f;
    wintab[i] = make_float4(w.x * sc, w.y * sc, w.z * sc, w.w * sc);
  }
  stage_tables4k<true>(p.tw, p.tw4k, tab, t4);
  const LdsTwiddles<true> tw = {tab, lane};
  const long long run = (long long)blockIdx.x * W4K + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long T = p.T;
  const long long c0 = LEAD + r * p.blocks_per_run;      // output hops q = 0 .. T - 2 are blocks c = q + LEAD
  long long c1 = c0 + p.blocks_per_run;
  if (c1 > LEAD + T - 1) c1 = LEAD + T - 1;
  if (c0 >= c1) return;
  const float4* env4 = reinterpret_cast<const float4*>(p.env);
  // reciprocal of the fully overlapped envelope: one division per wave instead of eight per hop (<= 1 ulp from acc / e)
  float4 rcpa[HS], rcpb[HS];
#pragma unroll
  for (int j = 0; j < HS; ++j) {
    const float4 ea = env4[(size_t)((1 << R) - 1) * (HOP / 4) + 2 * (lane + 64 * j)];
    const float4 eb = env4[(size_t)((1 << R) - 1) * (HOP / 4) + 2 * (lane + 64 * j) + 1];
    rcpa[j] = make_float4(1.0f / ea.x, 1.0f / ea.y, 1.0f / ea.z, 1.0f / ea.w);
    rcpb[j] = make_float4(1.0f / eb.x, 1.0f / eb.y, 1.0f / eb.z, 1.0f / eb.w);
  }
  float4 acca[8], accb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acca[j] = accb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float* yclip = p.y + b * (HOP * (T - 1));

  for (long long t = c0 - (R - 1); t < c1; ++t) {
    if (t >= 0 && t < T) {
      v2f y0[8], y1[8], y2[8], y3[8];
      spectrum_to_subffts4k<POLAR>(p.X, p.mag, p.phase, b * T + t, lane, w4, wr, y0, y1, y2, y3);
      fft512<true>(y0, tw, lds, lane);
      fft512<true>(y1, tw, lds, lane);
      fft512<true>(y2, tw, lds, lane);
      fft512<true>(y3, tw, lds, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 wa = wintab[2 * (lane + 64 * j)];
        const float4 wb = wintab[2 * (lane + 64 * j) + 1];
        acca[j].x += y0[j].x * wa.x;
        acca[j].y += y0[j].y * wa.y;
        acca[j].z += y1[j].x * wa.z;
        acca[j].w += y1[j].y * wa.w;
        accb[j].x += y2[j].x * wb.x;
        accb[j].y += y2[j].y * wb.y;
        accb[j].z += y3[j].x * wb.z;
        accb[j].w += y3[j].y * wb.w;
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
      if (mask == (1 << R) - 1) {       // every fully overlapped hop takes the reciprocal form, whichever run emits it
#pragma unroll
        for (int j = 0; j < HS; ++j) {
          v4f* d4 = reinterpret_cast<v4f*>(dst + 8 * (lane + 64 * j));
          __builtin_nontemporal_store((v4f){acca[j].x * rcpa[j].x, acca[j].y * rcpa[j].y, acca[j].z * rcpa[j].z,
                                            acca[j].w * rcpa[j].w}, d4);
          __builtin_nontemporal_store((v4f){accb[j].x * rcpb[j].x, accb[j].y * rcpb[j].y, accb[j].z * rcpb[j].z,
                                            accb[j].w * rcpb[j].w}, d4 + 1);
        }
      } else {
#pragma unroll
        for (int j = 0; j < HS; ++j) {
          const float4 ea = env4[(size_t)mask * (HOP / 4) + 2 * (lane + 64 * j)];
          const float4 eb = env4[(size_t)mask * (HOP / 4) + 2 * (lane + 64 * j) + 1];
          float4* d4 = reinterpret_cast<float4*>(dst + 8 * (lane + 64 * j));
          d4[0] = make_float4(acca[j].x / ea.x, acca[j].y / ea.y, acca[j].z / ea.z, acca[j].w / ea.w);
          d4[1] = make_float4(accb[j].x / eb.x, accb[j].y / eb.y, accb[j].z / eb.z, accb[j].w / eb.w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8 - HS; ++j) {
      acca[j] = acca[j + HS];
      accb[j] = accb[j + HS];
    }
#pragma unroll
    for (int j = 8 - HS; j < 8; ++j) acca[j] = accb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

int launch_istft4096_ola(const float2* X, const float* mag, const float* phase, long long B, long long T, int hop,
                         const float* window, const float* env, const float2* tw, const float2* tw4k, float* y,
                         hipStream_t stream) {
  if (B == 0 || T <= 1) return 0;
  P4kOla p = {X, mag, phase, window, env, tw, tw4k, y, B, T, 0, 0};
  const long long blocks = T - 1;                        // output hops per clip
  long long runs = (B >= 2048) ? 1 : (2048 + B - 1) / B;
  long long per = (blocks + runs - 1) / runs;
  const long long min_per = 8;
  if (per < min_per) per = min_per < blocks ? min_per : blocks;
  runs = (blocks + per - 1) / per;
  p.runs_per_clip = runs;
  p.blocks_per_run = per;
  const long long waves = B * runs;
  const unsigned grid = (unsigned)((waves + W4K - 1) / W4K);
#define OLA4K(POLAR_, HS_) hipLaunchKernelGGL((istft4096_ola_kernel<POLAR_, HS_>), dim3(grid), dim3(64 * W4K), 0, stream, p)
  const bool polar = (X == nullptr);
  if (hop == 512) { if (polar) OLA4K(true, 1); else OLA4K(false, 1); }
  else if (hop == 1024) { if (polar) OLA4K(true, 2); else OLA4K(false, 2); }
  else if (hop == 2048) { if (polar) OLA4K(true, 4); else OLA4K(false, 4); }
  else return -2;
#undef OLA4K
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

static long long frames_per_block_4k(long long nframes) {
  const long long max_blocks = 256LL * 8;
  long long fpb = (nframes + max_blocks - 1) / max_blocks;
  fpb = ((fpb + W4K - 1) / W4K) * W4K;
  return fpb < W4K ? W4K : fpb;
}

int launch_stft4096_fwd(const float* x, long long B, long long L, long long clip_stride, long long T, int hop, int center,
                        const float* window, const float2* tw, const float2* tw4k, float2* out, float* phase,
                        hipStream_t stream) {
  const long long nframes = B * T;
  if (nframes == 0) return 0;
  P4k p = {};
  p.x = x; p.window = window; p.tw = tw; p.tw4k = tw4k; p.X = out; p.phase_out = phase;
  p.L = L; p.clip_stride = clip_stride; p.T = T; p.total_frames = nframes; p.hop = hop; p.center = center;
  // the sliding-window / aligned-stream kernel: torch.stft's framing at hop n/4, 16-byte aligned clips, a 512-byte
  // aligned output, no phase side output
  if (center && hop == 1024 && !phase && L >= 4096 && (clip_stride & 3) == 0 && (((uintptr_t)x) & 15) == 0 &&
      (((uintptr_t)out) & 511) == 0 && (((uintptr_t)window) & 15) == 0 && variant(kVarFrameKernels) == 0) {
    P4kRun q = {};
    q.x = x; q.window = window; q.tw = tw; q.tw4k = tw4k; q.X = out;
    q.B = B; q.L = L; q.clip_stride = clip_stride; q.T = T;
    const long long slots = resident_waves(stft4096_run_fwd_kernel, 64 * W4K, 0);
    q.frames_per_run = plan_units_per_run(B, T, slots, 8, 1);
    q.runs_per_clip = (T + q.frames_per_run - 1) / q.frames_per_run;
    const long long waves = B * q.runs_per_clip;
    hipLaunchKernelGGL(stft4096_run_fwd_kernel, dim3((unsigned)((waves + W4K - 1) / W4K)), dim3(64 * W4K), 0, stream, q);
    return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  p.frames_per_block = frames_per_block_4k(nframes);
  const long long blocks = (nframes + p.frames_per_block - 1) / p.frames_per_block;
  if (phase) hipLaunchKernelGGL(stft4096_fwd_kernel<true>, dim3((unsigned)blocks), dim3(64 * W4K), 0, stream, p);
  else hipLaunchKernelGGL(stft4096_fwd_kernel<false>, dim3((unsigned)blocks), dim3(64 * W4K), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft4096_frames(const float2* X, const float* mag, const float* phase, long long nframes, const float* window,
                            const float2* tw, const float2* tw4k, float* frames, hipStream_t stream) {
  if (nframes == 0) return 0;
  P4k p = {};
  p.X = const_cast<float2*>(X); p.mag = mag; p.phase = phase; p.window = window; p.tw = tw; p.tw4k = tw4k; p.y = frames;
  p.total_frames = nframes;
  p.frames_per_block = frames_per_block_4k(nframes);
  const long long blocks = (nframes + p.frames_per_block - 1) / p.frames_per_block;
  if (X) hipLaunchKernelGGL(irfft4096_frames_kernel<false>, dim3((unsigned)blocks), dim3(64 * W4K), 0, stream, p);
  else hipLaunchKernelGGL(irfft4096_frames_kernel<true>, dim3((unsigned)blocks), dim3(64 * W4K), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
