// stft512.hip -- n_fft = 512 on the one-wavefront register FFT core (fft512.h): TWO frames per 512-point complex FFT.
//
// Replaces, for n_fft = 512 (any hop):  torch.stft(...).transpose(-2,-1)  (reference transforms/stft.py:98-104,
// dgt.py:64-70), rfft(x*window) on frames (stft.py:249-253, dgt.py:285-289), irfft(X)*inv_window (stft.py:260-266,
// dgt.py:296-302; the frames of torch.istft, overlap-added by stft_generic.hip's gather).  Until round 2 this size
// ran on the workgroup-per-frame LDS Stockham kernel of stft_generic.hip.
//
// A 512-point real transform is a 256-point complex FFT of a[n] = x[2n] + i x[2n+1] plus the real split.  Two
// frames A, B (consecutive frame indices) share one 512-point FFT: with y[2n] = a[n], y[2n+1] = b[n],
//   Y[k] = A[k] + W512^k B[k],   Y[k+256] = A[k] - W512^k B[k]          (k = 0 .. 255)
// so A[k] = (Y[k] + Y[k+256]) / 2 and B[k] = (Y[k] - Y[k+256]) conj(W512^k) / 2 -- lane-local, because the FFT
// leaves Y[lane + 64 m] in register m and k + 256 is register m + 4 of the same lane.  Even lanes load frame A, odd
// lanes frame B (y[lane + 64 j] = a or b [(lane >> 1) + 32 j]).  Then the real split of each,
//   X[k] = (A[k] + conj A[256-k])/2 - (i/2) W512^k (A[k] - conj A[256-k]),  k = 0 .. 256,
// with the mirror partner in lane 64 - lane, register 3 - m.  The inverse runs the same steps backwards.
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

constexpr int N5 = 512;
constexpr int F5 = N5 / 2 + 1;      // 257
constexpr int W5 = 4;               // waves per block

struct P5 {
  const float* x;
  const float* window;   // 512 samples
  const float2* tw;      // fft512 twiddle table
  const float2* tw512;   // W512^k, k = 0 .. 255
  float2* X;             // (frames, 257)
  const float* mag;
  const float* phase;
  float* phase_out;
  float* y;              // inverse: (frames, 512)
  long long L, clip_stride, T, total_frames, pairs_per_block;
  int hop, center;
};

__device__ __forceinline__ long long reflect5(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// this lane's 8 complex samples of ITS frame (frame f: even lanes of the pair's first frame, odd lanes the second):
// q[j] = (x[s + 2 n], x[s + 2 n + 1]), n = (lane >> 1) + 32 j; frames past the end read as zeros
__device__ __forceinline__ void load_half_frame5(const P5& p, long long f, int lane, float2 (&q)[8]) {
  if (f >= p.total_frames) {
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = make_float2(0.f, 0.f);
    return;
  }
  const long long b = f / p.T, t = f - b * p.T;
  const float* clip = p.x + b * p.clip_stride;
  const long long start = t * (long long)p.hop - (p.center ? N5 / 2 : 0);
  const bool interior = (start >= 0) && (start + N5 <= p.L);
  const int u = lane >> 1;
  if (interior && ((((uintptr_t)(clip + start)) & 7) == 0)) {
    const float2* src = reinterpret_cast<const float2*>(clip + start);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = src[u + 32 * j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long i0 = start + 2 * (u + 32 * j);
      float v[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const long long i = i0 + c;
        if (interior) v[c] = clip[i];
        else if (p.center) v[c] = clip[reflect5(i, p.L)];
        else v[c] = (i >= 0 && i < p.L) ? clip[i] : 0.0f;     // zero padding past the end (utils/misc.py:156)
      }
      q[j] = make_float2(v[0], v[1]);
    }
  }
}

// mirror partners P[m] = A[(256 - (lane + 64 m)) mod 256], m = 0 .. 3
__device__ __forceinline__ void mirror256(const v2f (&v)[4], v2f (&p)[4], int lane) {
  const int src = (64 - lane) & 63;
  v2f q[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const v2f a = q[3 - m];
    const v2f b = q[(4 - m) & 3];
    p[m] = (lane == 0) ? b : a;
  }
}

template <bool WRITE_PHASE>
__global__ __launch_bounds__(64 * W5) void stft512_fwd_kernel(P5 p) {
  __shared__ float2 lds_all[W5 * kFftLdsFloat2PerWave];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  Twiddles tw;
  load_twiddles<false>(tw, p.tw, lane);
  v2f w5[4];                                   // W512^k, k = lane + 64 m
  float2 win[8];                               // this lane's window samples (same for frame A and B lanes)
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float2 a = p.tw512[lane + 64 * m];
    w5[m] = (v2f){a.x, a.y};
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) win[j] = reinterpret_cast<const float2*>(p.window)[(lane >> 1) + 32 * j];
  const long long n_pairs = (p.total_frames + 1) / 2;
  const long long pr_begin = (long long)blockIdx.x * p.pairs_per_block;
  long long pr_end = pr_begin + p.pairs_per_block;
  if (pr_end > n_pairs) pr_end = n_pairs;
  const v2f hh = {0.5f, 0.5f};

  long long pr = pr_begin + wave;
  float2 nxt[8];
  if (pr < pr_end) load_half_frame5(p, 2 * pr + (lane & 1), lane, nxt);
  for (; pr < pr_end; pr += W5) {
    v2f y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (v2f){nxt[j].x * win[j].x, nxt[j].y * win[j].y};
    if (pr + W5 < pr_end) load_half_frame5(p, 2 * (pr + W5) + (lane & 1), lane, nxt);
    fft512<false>(y, tw, lds, lane);
    // unpack the two 256-point spectra (halved: the real split wants A/2): HA = (Y[k] + Y[k+256]) / 4 ...
    v2f ha[4], hb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const v2f s = (y[m] + y[m + 4]) * hh;              // A[k]
      const v2f d = (y[m] - y[m + 4]) * hh;              // W512^k B[k]
      ha[m] = s * hh;                                     // A / 2
      hb[m] = cmul_conj_v(d, w5[m]) * hh;                 // B / 2
    }
    v2f pa[4], pb[4];
    mirror256(ha, pa, lane);
    mirror256(hb, pb, lane);
    const long long fa = 2 * pr, fb = 2 * pr + 1;
    float2* rowa = p.X + fa * F5;
    float2* rowb = rowa + F5;
    const bool has_b = fb < p.total_frames;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const v2f xa = add_mi(add_conj(ha[m], pa[m]), cmul_v(sub_conj(ha[m], pa[m]), w5[m]));
      const v2f xb = add_mi(add_conj(hb[m], pb[m]), cmul_v(sub_conj(hb[m], pb[m]), w5[m]));
      rowa[lane + 64 * m] = to_f2(xa);
      if (has_b) rowb[lane + 64 * m] = to_f2(xb);
      if (WRITE_PHASE) {
        p.phase_out[fa * F5 + lane + 64 * m] = fast_atan2f(xa.y, xa.x);
        if (has_b) p.phase_out[fb * F5 + lane + 64 * m] = fast_atan2f(xb.y, xb.x);
      }
    }
    if (lane == 0) {
      const float na = 2.0f * (ha[0].x - ha[0].y), nb = 2.0f * (hb[0].x - hb[0].y);   // X[256] = Re A[0] - Im A[0]
      rowa[256] = make_float2(na, 0.0f);
      if (has_b) rowb[256] = make_float2(nb, 0.0f);
      if (WRITE_PHASE) {
        p.phase_out[fa * F5 + 256] = fast_atan2f(0.0f, na);
        if (has_b) p.phase_out[fb * F5 + 256] = fast_atan2f(0.0f, nb);
      }
    }
  }
}


// ---------------------------------------------------------------------------
// forward, hop = 128 = N/4, center = True: sliding window in registers + aligned stream stores (round 3; the scheme of
// stft1024.hip's AL kernel).  A wave walks consecutive frame PAIRS (2 i, 2 i + 1) of one clip; even lanes hold the
// samples of the first frame, odd lanes of the second (register slot j = 64 samples), so the next pair is "slots j + 4
// of the same lane": four 8-byte loads per lane and pair instead of eight.  Rows are 2056 bytes (8 f bytes past a
// 128-byte line).  The FFT's output columns are rotated over the lanes by rot = (f 257) mod 64 of the pair's FIRST
// frame; after the un-mixing every lane holds the same columns of both frames, so the second frame -- which needs
// rot + 1 -- is moved up one lane (nine ds_bpermute: its four registers and its Nyquist bin) and then leaves exactly
// like a frame of its own: four full-line non-temporal stores per frame, block 4 (the tail of register 3, then the
// Nyquist bin) carried into the next frame's block 0.
// ---------------------------------------------------------------------------
struct P5Run {
  const float* x;
  const float* window;
  const float2* tw;
  const float2* tw512;
  float2* X;
  long long B, L, clip_stride, T, runs_per_clip, pairs_per_run;
};

__device__ __forceinline__ void mirror256_rot(const v2f (&v)[4], v2f (&p)[4], int lane, int rot, int col) {
  const int src = (2 * rot - lane) & 63;
  v2f q[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const v2f a = q[3 - m];
    const v2f b = q[(4 - m) & 3];
    p[m] = (col == 0) ? b : a;
  }
}

// two samples of a frame that starts at original index s (reflect-padded)
__device__ __forceinline__ float2 load_pair5(const float* clip, long long L, long long i, bool interior) {
  if (interior) return *reinterpret_cast<const float2*>(clip + i);        // clip base 8-byte aligned, i even
  return make_float2(clip[reflect5(i, L)], clip[reflect5(i + 1, L)]);
}

constexpr int W5R = 8;      // waves per block of the run kernel

__global__ __launch_bounds__(64 * W5R, 4) void stft512_run_fwd_kernel(P5Run p) {
  __shared__ float2 lds_all[W5R * kFftLdsFloat2PerWave + 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int par = lane & 1, u = lane >> 1;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* w5tab = lds_all + W5R * kFftLdsFloat2PerWave;                  // W512^k, k = 0 .. 255
  for (int i = threadIdx.x; i < 256; i += 64 * W5R) w5tab[i] = p.tw512[i];
  __syncthreads();
  Twiddles tw;
  load_twiddles<false>(tw, p.tw, lane);
  float2 win[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) win[j] = reinterpret_cast<const float2*>(p.window)[u + 32 * j];

  const long long run = (long long)blockIdx.x * W5R + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long T = p.T, L = p.L;
  const long long n_pairs = (T + 1) / 2;
  const long long i0 = r * p.pairs_per_run;
  long long i1 = i0 + p.pairs_per_run;
  if (i1 > n_pairs) i1 = n_pairs;
  if (i0 >= i1) return;
  const float* clip = p.x + b * p.clip_stride;
  const v2f hh = {0.5f, 0.5f};

  // this lane's frame of pair i: f = 2 i + par, starting at original sample 128 f - 256; slot j = samples 64 j .. + 63
  float2 raw[8];
  {
    const long long f = 2 * i0 + par;
    const long long s = 128 * f - 256;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long i = s + 2 * (u + 32 * j);
      const bool in = (s + 64 * j >= 0) && (s + 64 * j + 64 <= L);
      raw[j] = (f < T) ? load_pair5(clip, L, i, in) : make_float2(0.f, 0.f);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(raw[j].x), "+v"(raw[j].y));

  const long long e0 = (b * T + 2 * i0) * F5;
  int rot = (int)(e0 & 63);
  float2* sp = p.X + (e0 - rot) + lane;
  v2f carry = {0.f, 0.f};
  bool head = true;
  auto put = [&](float2* dst, v2f val) { __builtin_nontemporal_store(val, reinterpret_cast<v2f*>(dst)); };
  // one frame's four registers (columns already rotated by the current `rot`) and Nyquist bin (on lane rot) into the stream
  auto emit_frame = [&](const v2f (&z)[4], v2f nyq) {
    const bool lo = lane < rot;
    const v2f s0 = lo ? carry : z[0];
    if (head) {
      if (!lo) put(sp, s0);
      head = false;
    } else {
      put(sp, s0);
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) put(sp + 64 * j, lo ? z[j - 1] : z[j]);
    carry = lo ? z[3] : nyq;
    if (rot == 63) {
      put(sp + 256, carry);
      sp += 320;
      rot = 0;
    } else {
      sp += 256;
      ++rot;
    }
  };

  auto pair_body = [&](const float2 (&fresh)[4], bool has_b) {
    wave_priority<3>();        // transform > stores, as in stft1024.hip
    v2f y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (v2f){raw[j].x * win[j].x, raw[j].y * win[j].y};
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = raw[j + 4];
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[4 + j] = fresh[j];
    const int col = (lane - rot) & 63;
    fft512<false>(y, tw, lds, lane, col);
    const v2f* w5 = reinterpret_cast<const v2f*>(w5tab) + col;             // w5[64 m] = W512^(col + 64 m)
    v2f ha[4], hb[4], wk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      wk[m] = lds_read_single(w5 + 64 * m);
      const v2f sm = (y[m] + y[m + 4]) * hh;              // A[k]
      const v2f d = (y[m] - y[m + 4]) * hh;               // W512^k B[k]
      ha[m] = sm * hh;                                     // A / 2
      hb[m] = cmul_conj_v(d, wk[m]) * hh;                  // B / 2
    }
    v2f pa[4], pb[4];
    mirror256_rot(ha, pa, lane, rot, col);
    mirror256_rot(hb, pb, lane, rot, col);
    v2f xa[4], xb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      xa[m] = add_mi(add_conj(ha[m], pa[m]), cmul_v(sub_conj(ha[m], pa[m]), wk[m]));
      xb[m] = add_mi(add_conj(hb[m], pb[m]), cmul_v(sub_conj(hb[m], pb[m]), wk[m]));
    }
    const v2f na = {2.0f * (ha[0].x - ha[0].y), 0.0f};     // X_A[256], on the lane whose column is 0 (= lane rot)
    const float nbx = 2.0f * (hb[0].x - hb[0].y);
    wave_priority<1>();
    emit_frame(xa, na);
    if (has_b) {
      // the second frame one lane up: its columns then sit where a frame with rotation rot + 1 wants them
      const int src = (lane - 1) & 63;
      v2f xs[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        xs[m].x = __shfl(xb[m].x, src, 64);
        xs[m].y = __shfl(xb[m].y, src, 64);
      }
      const v2f nb = {__shfl(nbx, src, 64), 0.0f};
      emit_frame(xs, nb);
    }
    wave_priority<0>();
  };

  long long i = i0;
  // pairs whose successor pair (frames 2 i + 2, 2 i + 3) exists entirely and takes its new samples from inside the clip
  long long i_fast_end = 0;
  if (L >= 256 + 128 * 3) {
    // frame f's new samples are [128 f, 128 f + 256): need 128 (2 i + 3) + 256 <= L and 2 i + 3 <= T - 1
    long long lim = (L - 256) / 128;                 // largest f with 128 f + 256 <= L
    if (lim > T - 1) lim = T - 1;
    i_fast_end = (lim - 3) / 2 + 1;                  // pairs i with 2 i + 3 <= lim
    if (lim < 3) i_fast_end = 0;
  }
  if (i_fast_end > i1 - 1) i_fast_end = i1 - 1;
  if (i < i_fast_end) {
    const float2* nsrc = reinterpret_cast<const float2*>(clip + 128 * (2 * (i + 1) + par)) + u;
    for (; i < i_fast_end; ++i) {
      float2 fresh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fresh[j] = nsrc[32 * j];
      nsrc += 128;                                    // two hops = 256 samples
      pair_body(fresh, true);
    }
  }
  for (; i < i1; ++i) {
    float2 fresh[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    if (i + 1 < i1) {
      const long long f = 2 * (i + 1) + par;
      if (f < T) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long long s = 128 * f + 64 * j;       // slot 4 + j of frame f: samples 128 f - 256 + 64 (4 + j) ..
          fresh[j] = load_pair5(clip, L, s + 2 * u, s >= 0 && s + 64 <= L);
        }
      }
    }
    pair_body(fresh, 2 * i + 1 < T);
  }
  if (lane < rot) put(sp, carry);
}

__device__ __forceinline__ void sincos_big5(float phase, float& s, float& c) {
  double t = (double)phase * 0.15915494309189533577;  // 1 / (2 pi)
  t -= rint(t);
  const float r = (float)t;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

// one frame's one-sided spectrum -> A[k] * 2 (k = lane + 64 m, m = 0..3): E + i O of the inverse split
template <bool POLAR>
__device__ __forceinline__ void load_split5(const P5& p, long long f, bool exists, int lane, const v2f (&w5)[4],
                                            v2f (&a)[4]) {
  if (!exists) {              // wave-uniform: a missing frame contributes a zero spectrum
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = (v2f){0.f, 0.f};
    return;
  }
  v2f v[4];
  float nyq_re;
  if (POLAR) {
    const float* mrow = p.mag + f * F5;
    const float* prow = p.phase + f * F5;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float sn, cs;
      const float g = mrow[lane + 64 * m];
      sincos_big5(prow[lane + 64 * m], sn, cs);
      v[m] = (v2f){g * cs, g * sn};
    }
    float sn, cs;
    sincos_big5(prow[256], sn, cs);
    nyq_re = mrow[256] * cs;
  } else {
    const float2* row = p.X + f * F5;
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = to_v(row[lane + 64 * m]);
    nyq_re = row[256].x;
  }
  if (lane == 0) v[0].y = 0.0f;                 // c2r ignores the imaginary parts of DC and Nyquist
  v2f pm[4];
  mirror256(v, pm, lane);
  if (lane == 0) pm[0] = (v2f){nyq_re, 0.0f};   // partner of k = 0 is X[256]
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const v2f e = add_conj(v[m], pm[m]);
    const v2f d = cmul_conj_v(sub_conj(v[m], pm[m]), w5[m]);
    a[m] = add_pi(e, d);                         // 2 A[k]
  }
}

// irfft(X) * window for frame pairs, frames out (overlap-add: stft_generic.hip's gather)
template <bool POLAR>
__global__ __launch_bounds__(64 * W5) void irfft512_frames_kernel(P5 p) {
  __shared__ float2 lds_all[W5 * kFftLdsFloat2PerWave];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  Twiddles tw;
  load_twiddles<true>(tw, p.tw, lane);
  v2f w5[4];
  float2 win[8];
#pragma unroll
  for (int m = 0; m < 4; ++m) w5[m] = to_v(p.tw512[lane + 64 * m]);
  const float scale = 1.0f / 1024.0f;           // 1 / 512 of the transform, 1 / 2 of the split (E + i O = 2 A)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float2 w = reinterpret_cast<const float2*>(p.window)[(lane >> 1) + 32 * j];
    win[j] = make_float2(w.x * scale, w.y * scale);
  }
  const long long n_pairs = (p.total_frames + 1) / 2;
  const long long pr_begin = (long long)blockIdx.x * p.pairs_per_block;
  long long pr_end = pr_begin + p.pairs_per_block;
  if (pr_end > n_pairs) pr_end = n_pairs;
  for (long long pr = pr_begin + wave; pr < pr_end; pr += W5) {
    v2f a[4], b[4];
    load_split5<POLAR>(p, 2 * pr, 2 * pr < p.total_frames, lane, w5, a);
    load_split5<POLAR>(p, 2 * pr + 1, 2 * pr + 1 < p.total_frames, lane, w5, b);
    // Y[k] = A + W512^k B, Y[k+256] = A - W512^k B
    v2f y[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const v2f t = cmul_v(b[m], w5[m]);
      y[m] = a[m] + t;
      y[m + 4] = a[m] - t;
    }
    fft512<true>(y, tw, lds, lane);
    // y[lane + 64 j]: even lanes hold frame A's complex sample (lane >> 1) + 32 j, odd lanes frame B's
    const long long f = 2 * pr + (lane & 1);
    if (f < p.total_frames) {
      float2* dst = reinterpret_cast<float2*>(p.y + f * N5);
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[(lane >> 1) + 32 * j] = make_float2(y[j].x * win[j].x, y[j].y * win[j].y);
    }
  }
}

// ---------------------------------------------------------------------------
// torch.istft for n_fft = 512, hop = 64 / 128 / 256 in one kernel: irfft + window + overlap-add + envelope
// (reference stft.py:120-128, dgt.py:86-93; polar input stft.py:157-161, dgt.py:152-154).
// A wave walks consecutive frame PAIRS (t, t + 1) of one clip; even lanes hold frame t, odd lanes frame t + 1, the
// same complex sample index (lane >> 1) + 32 j in register slot j (a slot = 64 samples, a hop = HS slots).  Every lane
// overlap-adds the frames of ITS parity in registers (they are two hops apart: the window slides 2 HS slots per pair
// and the HS slots behind the first hop are kept as `carry` when they slide out).  With the pair's frames added,
//   block t     = even[0 .. HS)      + odd.carry      (odd frames t - 1, t - 3, ... : their second hop)
//   block t + 1 = even[HS .. 2 HS)   + odd[0 .. HS)
// are complete: one DPP exchange with the neighbouring lane adds the two parities, the sum is divided by the window
// envelope of the frames that exist around the block (2^R x hop table of at_istft_envelope_table) and stored once.
// Output sample s is padded sample s + 256: block c is output hop c - 256 / hop.
// ---------------------------------------------------------------------------
struct P5Ola {
  const float2* X;
  const float* mag;
  const float* phase;
  const float* window;
  const float* env;      // 2^R x hop
  const float2* tw;
  const float2* tw512;
  float* y;              // (B, hop (T - 1))
  long long B, T, runs_per_clip, pairs_per_run;
};

__device__ __forceinline__ v2f dpp_xor1_v2(v2f a) {
  v2f r;
  r.x = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(a.x), 0xB1, 0xF, 0xF, true));
  r.y = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(a.y), 0xB1, 0xF, 0xF, true));
  return r;
}

template <bool POLAR, int HS>
__global__ __launch_bounds__(64 * W5) void istft512_ola_kernel(P5Ola p) {
  constexpr int HOP = 64 * HS, R = 8 / HS, LEAD = 256 / HOP;
  __shared__ float2 lds_all[W5 * kFftLdsFloat2PerWave];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  const int par = lane & 1, u = lane >> 1;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  Twiddles tw;
  load_twiddles<true>(tw, p.tw, lane);
  v2f w5[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) w5[m] = to_v(p.tw512[lane + 64 * m]);
  const float scale = 1.0f / 1024.0f;
  float2 win[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float2 w = reinterpret_cast<const float2*>(p.window)[u + 32 * j];
    win[j] = make_float2(w.x * scale, w.y * scale);
  }
  const long long run = (long long)blockIdx.x * W5 + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long T = p.T;
  // output hops q = 0 .. T - 2 are blocks c = q + LEAD; pair i holds frames 2 i, 2 i + 1 and completes blocks 2 i, 2 i + 1
  const long long c_lo = LEAD, c_hi = LEAD + T - 1;                       // valid blocks [c_lo, c_hi)
  const long long i0 = c_lo / 2 + r * p.pairs_per_run;
  long long i1 = i0 + p.pairs_per_run;
  const long long i_end = (c_hi + 1) / 2;                                  // first pair with no valid block
  if (i1 > i_end) i1 = i_end;
  if (i0 >= i1) return;
  // P5 view for the shared spectrum loader
  P5 q5 = {};
  q5.X = const_cast<float2*>(p.X); q5.mag = p.mag; q5.phase = p.phase;
  const float2* env2 = reinterpret_cast<const float2*>(p.env);
  float* yclip = p.y + b * (HOP * (T - 1));
  v2f acc[8], carry[HS];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (v2f){0.f, 0.f};
#pragma unroll
  for (int j = 0; j < HS; ++j) carry[j] = (v2f){0.f, 0.f};

  // reciprocal of the fully overlapped envelope, once per wave: in the steady state a block needs neither the envelope
  // load nor the division (<= 1 ulp from sum / e); every fully overlapped block takes this form, whichever run emits it
  float2 rcp_full[HS];
#pragma unroll
  for (int j = 0; j < HS; ++j) {
    const float2 e = env2[(size_t)((1 << R) - 1) * (HOP / 2) + u + 32 * j];
    rcp_full[j] = make_float2(1.0f / e.x, 1.0f / e.y);
  }
  constexpr int WARM = (R + 1) / 2;                                       // pairs that start early: R - 1 frames back
  for (long long i = i0 - WARM; i < i1; ++i) {
    // slide this lane's window two hops; what leaves behind the first hop is the carry
#pragma unroll
    for (int j = 0; j < HS; ++j) carry[j] = (HS + j < 8) ? acc[HS + j] : (v2f){0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (j + 2 * HS < 8) ? acc[j + 2 * HS] : (v2f){0.f, 0.f};
    const long long ta = 2 * i, tb = 2 * i + 1;
    if (tb >= 0 && ta < T) {                     // at least one frame of the pair exists (wave-uniform)
      v2f a[4], bb[4];                           // frames outside [0, T) of THIS clip read as zero spectra
      load_split5<POLAR>(q5, b * T + ta, ta >= 0 && ta < T, lane, w5, a);
      load_split5<POLAR>(q5, b * T + tb, tb >= 0 && tb < T, lane, w5, bb);
      v2f y[8];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const v2f t = cmul_v(bb[m], w5[m]);
        y[m] = a[m] + t;
        y[m + 4] = a[m] - t;
      }
      fft512<true>(y, tw, lds, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (v2f){y[j].x * win[j].x, y[j].y * win[j].y};
    }
    if (i < i0) continue;
    // the two blocks this pair completes
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const long long c = 2 * i + half;
      int mask = 0;
#pragma unroll
      for (int qq = 0; qq < R; ++qq) {
        const long long ft = c - (R - 1) + qq;
        if (ft >= 0 && ft < T) mask |= 1 << qq;
      }
      const bool valid = c >= c_lo && c < c_hi;
      float2* dst = reinterpret_cast<float2*>(yclip + (c - LEAD) * HOP);
#pragma unroll
      for (int j = 0; j < HS; ++j) {
        // this lane's parity share of slot j of the block, then the neighbour's
        const v2f mine = (half == 0) ? (par ? carry[j] : acc[j]) : (par ? acc[j] : acc[HS + j]);
        const v2f sum = mine + dpp_xor1_v2(mine);
        // a lane pair holds the same sum: even lanes store the first half of the slot pair range, odd lanes ... both
        // lanes of a pair would write the same 8 bytes, so the even lane alone stores
        if (valid && par == 0) {
          if (mask == (1 << R) - 1) {
            __builtin_nontemporal_store((v2f){sum.x * rcp_full[j].x, sum.y * rcp_full[j].y},
                                        reinterpret_cast<v2f*>(dst + u + 32 * j));
          } else {
            const float2 e = env2[(size_t)mask * (HOP / 2) + u + 32 * j];
            dst[u + 32 * j] = make_float2(sum.x / e.x, sum.y / e.y);
          }
        }
      }
    }
  }
}

int launch_istft512_ola(const float2* X, const float* mag, const float* phase, long long B, long long T, int hop,
                        const float* window, const float* env, const float2* tw, const float2* tw512, float* y,
                        hipStream_t stream) {
  if (B == 0 || T <= 1) return 0;
  P5Ola p = {X, mag, phase, window, env, tw, tw512, y, B, T, 0, 0};
  const long long lead = 256 / hop;
  const long long pairs = (lead + T - 1 + 1) / 2 - lead / 2;             // pairs that hold a valid block
  long long runs = (B >= 4096) ? 1 : (4096 + B - 1) / B;
  long long per = (pairs + runs - 1) / runs;
  const long long min_per = 16;
  if (per < min_per) per = min_per < pairs ? min_per : pairs;
  if (per < 1) per = 1;
  runs = (pairs + per - 1) / per;
  p.runs_per_clip = runs;
  p.pairs_per_run = per;
  const long long waves = B * runs;
  const unsigned grid = (unsigned)((waves + W5 - 1) / W5);
#define OLA5(POLAR_, HS_) hipLaunchKernelGGL((istft512_ola_kernel<POLAR_, HS_>), dim3(grid), dim3(64 * W5), 0, stream, p)
  const bool polar = (X == nullptr);
  if (hop == 64) { if (polar) OLA5(true, 1); else OLA5(false, 1); }
  else if (hop == 128) { if (polar) OLA5(true, 2); else OLA5(false, 2); }
  else if (hop == 256) { if (polar) OLA5(true, 4); else OLA5(false, 4); }
  else return -2;
#undef OLA5
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

static long long pairs_per_block_5(long long npairs) {
  const long long max_blocks = 256LL * 8;
  long long ppb = (npairs + max_blocks - 1) / max_blocks;
  ppb = ((ppb + W5 - 1) / W5) * W5;
  return ppb < W5 ? W5 : ppb;
}

// ---------------------------------------------------------------------------
// Features-only forward for n_fft = 512, any hop: audio -> normalise(contrast(|X|^p @ bank)) for a banded bank
// (MelSpectrogram / MFCC at the usual speech setting; mel.py:43-44, 68-73 behind stft.py:98-104).  The forward kernel
// with both spectra of a frame pair kept in registers: |X|^p of frames A and B go into two LDS rows of the wave and the
// band walk reads every weight quad once for both rows.  Output row-major, or channel-major through the eight-frame
// register window with sector-aligned flushes (mel_banded.hip).  A wave takes a run of consecutive frame pairs.
// ---------------------------------------------------------------------------
struct P5Mel {
  const float* x;
  const float* window;
  const float2* tw;
  const float2* tw512;
  float* feat;           // (B*T, N) or (B, N, T)
  const float* offset;
  const float* scale;
  long long L, clip_stride, T, total_frames, pairs_per_wave;
  BandBank bank;
  int hop, contrast, power2, channel_major, row_floats, table_floats;
  float eps;
};

__device__ __forceinline__ float contrast5(float v, int mode, float eps) {
  switch (mode) {
    case C_LOG1P: return logf(1.0f + v);
    case C_LOG: return logf(fmaxf(v, eps));
    case C_LOG10: return log10f(fmaxf(v, eps));
    default: return v;
  }
}

template <int CMW>    // 1 / 2: channel-major output of a bank with that many passes (register window); 0: anything else
__global__ __launch_bounds__(64 * W5) void stft512_mel_kernel(P5Mel p) {
  __shared__ float2 lds_all[W5 * kFftLdsFloat2PerWave];
  extern __shared__ __attribute__((aligned(16))) float dyn5[];      // rows (two per wave), weights, lane tables
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: run bookkeeping on the scalar unit
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float* rowa = dyn5 + (2 * wave) * p.row_floats;
  float* rowb = rowa + p.row_floats;
  float* wlds = dyn5 + 2 * W5 * p.row_floats;
  int* lane_tab = reinterpret_cast<int*>(wlds + p.table_floats);
  for (int i = threadIdx.x; i < p.table_floats; i += 64 * W5) wlds[i] = p.bank.weights[i];
  for (int i = threadIdx.x; i < 64 * p.bank.n_passes; i += 64 * W5) {
    lane_tab[i] = p.bank.lane_start[i];
    lane_tab[64 * p.bank.n_passes + i] = p.bank.lane_filter[i];
  }
  for (int k = 256 + lane; k < p.row_floats; k += 64) rowa[k] = rowb[k] = 0.0f;      // bin 256 is rewritten per frame
  __syncthreads();
  Twiddles tw;
  load_twiddles<false>(tw, p.tw, lane);
  v2f w5[4];
  float2 win[8];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float2 a = p.tw512[lane + 64 * m];
    w5[m] = (v2f){a.x, a.y};
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) win[j] = reinterpret_cast<const float2*>(p.window)[(lane >> 1) + 32 * j];
  const long long n_pairs = (p.total_frames + 1) / 2;
  const long long pr_begin = ((long long)blockIdx.x * W5 + wave) * p.pairs_per_wave;
  long long pr_end = pr_begin + p.pairs_per_wave;
  if (pr_end > n_pairs) pr_end = n_pairs;
  if (pr_begin >= pr_end) return;
  const v2f hh = {0.5f, 0.5f};
  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }
  P5 lp = {};
  lp.x = p.x; lp.L = p.L; lp.clip_stride = p.clip_stride; lp.T = p.T; lp.total_frames = p.total_frames; lp.hop = p.hop;
  lp.center = 1;
  long long f_last = 2 * pr_end - 1;                     // last frame of this wave's run
  if (f_last > p.total_frames - 1) f_last = p.total_frames - 1;

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
  long long cb = (2 * pr_begin) / p.T, ct = 2 * pr_begin - cb * p.T;
  int fq[CMW > 0 ? CMW : 1];
  if (CMW > 0) {
#pragma unroll
    for (int q = 0; q < (CMW > 0 ? CMW : 1); ++q) fq[q] = lane_tab[(CMW + q) * 64 + lane];
  }

  // one frame's features out: row-major / scalar channel-major, or through the window
  auto emit = [&](long long f, const float (&acc_q)[16], int n_acc) {
    const long long b = cb, t = ct;
    if (++ct == p.T) {
      ct = 0;
      ++cb;
    }
    if constexpr (CMW > 0) {
      const bool last_of_run = (t == p.T - 1) || (f == f_last);
#pragma unroll
      for (int q = 0; q < CMW; ++q) {
#pragma unroll
        for (int k = 0; k < 7; ++k) cm[q][k] = cm[q][k + 1];
        cm[q][7] = acc_q[q];
        ++held[q];
        if (fq[q] >= 0) {
          if (!e_valid) e_next[q] = (b * p.bank.n_filters + fq[q]) * p.T + t + 1;
          const long long e = e_next[q];
          e_next[q] = e + ((t == p.T - 1) ? (long long)(p.bank.n_filters - 1) * p.T + 1 : 1);
          if ((e & 7) == 0 || last_of_run) {
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
#pragma unroll
      for (int q = 0; q < 16; ++q) {                       // static indices into the accumulator registers
        if (q < n_acc) {
          const int filt = lane_tab[(p.bank.n_passes + q) * 64 + lane];
          if (filt >= 0) {
            if (p.channel_major) p.feat[(b * p.bank.n_filters + filt) * p.T + t] = acc_q[q];
            else p.feat[f * p.bank.n_filters + filt] = acc_q[q];
          }
        }
      }
    }
  };

  float2 nxt[8];
  load_half_frame5(lp, 2 * pr_begin + (lane & 1), lane, nxt);
  for (long long pr = pr_begin; pr < pr_end; ++pr) {
    wave_priority<3>();        // transform > epilogue, as in stft1024.hip
    v2f y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = (v2f){nxt[j].x * win[j].x, nxt[j].y * win[j].y};
    if (pr + 1 < pr_end) load_half_frame5(lp, 2 * (pr + 1) + (lane & 1), lane, nxt);
    fft512<false>(y, tw, lds, lane);
    v2f ha[4], hb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const v2f s = (y[m] + y[m + 4]) * hh;
      const v2f d = (y[m] - y[m + 4]) * hh;
      ha[m] = s * hh;
      hb[m] = cmul_conj_v(d, w5[m]) * hh;
    }
    v2f pa[4], pb[4];
    mirror256(ha, pa, lane);
    mirror256(hb, pb, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const v2f xa = add_mi(add_conj(ha[m], pa[m]), cmul_v(sub_conj(ha[m], pa[m]), w5[m]));
      const v2f xb = add_mi(add_conj(hb[m], pb[m]), cmul_v(sub_conj(hb[m], pb[m]), w5[m]));
      const float sa = fmaf(xa.x, xa.x, xa.y * xa.y), sb = fmaf(xb.x, xb.x, xb.y * xb.y);
      rowa[lane + 64 * m] = p.power2 ? sa : __builtin_amdgcn_sqrtf(sa);
      rowb[lane + 64 * m] = p.power2 ? sb : __builtin_amdgcn_sqrtf(sb);
    }
    if (lane == 0) {
      const float na = 2.0f * (ha[0].x - ha[0].y), nb = 2.0f * (hb[0].x - hb[0].y);
      rowa[256] = p.power2 ? na * na : fabsf(na);
      rowb[256] = p.power2 ? nb * nb : fabsf(nb);
    }
    wave_lds_sync();
    wave_priority<0>();
    // one walk for both frames: every weight quad is read once
    const float4* w = reinterpret_cast<const float4*>(wlds) + lane;
    constexpr int NQ = CMW > 0 ? CMW : 16;
    float fa_q[16], fb_q[16];
    const int n_acc = CMW > 0 ? CMW : p.bank.n_passes;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q < n_acc) {
        const int st = lane_tab[q * 64 + lane];
        const float4* a4 = reinterpret_cast<const float4*>(rowa + st);
        const float4* b4 = reinterpret_cast<const float4*>(rowb + st);
        v2f aa = {0.f, 0.f}, ab = {0.f, 0.f};
        const int quads = p.bank.pass_len[q] >> 2;
        int j = 0;
        for (; j + 2 <= quads; j += 2) {          // two steps' reads (six quads) ahead of the multiply-adds
          float4 wv[2], av[2], bv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            wv[u] = w[(j + u) * 64];
            av[u] = a4[j + u];
            bv[u] = b4[j + u];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            aa = __builtin_elementwise_fma((v2f){av[u].x, av[u].y}, (v2f){wv[u].x, wv[u].y}, aa);
            aa = __builtin_elementwise_fma((v2f){av[u].z, av[u].w}, (v2f){wv[u].z, wv[u].w}, aa);
            ab = __builtin_elementwise_fma((v2f){bv[u].x, bv[u].y}, (v2f){wv[u].x, wv[u].y}, ab);
            ab = __builtin_elementwise_fma((v2f){bv[u].z, bv[u].w}, (v2f){wv[u].z, wv[u].w}, ab);
          }
        }
        for (; j < quads; ++j) {
          const float4 wv = w[j * 64], av = a4[j], bv = b4[j];
          aa = __builtin_elementwise_fma((v2f){av.x, av.y}, (v2f){wv.x, wv.y}, aa);
          aa = __builtin_elementwise_fma((v2f){av.z, av.w}, (v2f){wv.z, wv.w}, aa);
          ab = __builtin_elementwise_fma((v2f){bv.x, bv.y}, (v2f){wv.x, wv.y}, ab);
          ab = __builtin_elementwise_fma((v2f){bv.z, bv.w}, (v2f){wv.z, wv.w}, ab);
        }
        w += quads * 64;
        float va = contrast5(aa.x + aa.y, p.contrast, p.eps), vb = contrast5(ab.x + ab.y, p.contrast, p.eps);
        if (p.offset) {
          va = (va - off) / sc;
          vb = (vb - off) / sc;
        }
        fa_q[q] = va;
        fb_q[q] = vb;
      }
    }
    emit(2 * pr, fa_q, n_acc);
    if (2 * pr + 1 < p.total_frames) emit(2 * pr + 1, fb_q, n_acc);
    wave_lds_sync();
  }
}

int launch_stft512_mel(const float* x, long long B, long long L, long long clip_stride, long long T, int hop,
                       const float* window, const float2* tw, const float2* tw512, const BandBank* bank, int contrast,
                       int power2, const float* offset, const float* scale, float eps, float* feat, int channel_major,
                       hipStream_t stream) {
  const long long nframes = B * T;
  if (nframes == 0) return 0;
  P5Mel p = {};
  p.x = x; p.window = window; p.tw = tw; p.tw512 = tw512; p.feat = feat; p.offset = offset; p.scale = scale;
  p.L = L; p.clip_stride = clip_stride; p.T = T; p.total_frames = nframes; p.bank = *bank; p.hop = hop;
  p.contrast = contrast; p.power2 = power2; p.channel_major = channel_major; p.eps = eps;
  int max_walk = 0, table_floats = 0;
  for (int q = 0; q < bank->n_passes; ++q) {
    max_walk = bank->pass_len[q] > max_walk ? bank->pass_len[q] : max_walk;
    table_floats += 64 * bank->pass_len[q];
  }
  p.table_floats = table_floats;
  p.row_floats = (F5 + max_walk + 63) / 64 * 64;
  const size_t lds = sizeof(float) * ((size_t)2 * W5 * p.row_floats + table_floats) + sizeof(int) * (size_t)2 * 64 * bank->n_passes;
  if (lds > 48 * 1024) return -2;
  const long long npairs = (nframes + 1) / 2;
  long long ppw = (npairs + 256LL * 8 * W5 - 1) / (256LL * 8 * W5);
  if (ppw < 4) ppw = 4;
  p.pairs_per_wave = ppw;
  const long long waves = (npairs + ppw - 1) / ppw;
  const unsigned grid = (unsigned)((waves + W5 - 1) / W5);
  void (*kernel)(P5Mel) = stft512_mel_kernel<0>;
  if (channel_major && bank->n_passes == 1) kernel = stft512_mel_kernel<1>;
  else if (channel_major && bank->n_passes == 2) kernel = stft512_mel_kernel<2>;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * W5), lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_stft512_fwd(const float* x, long long B, long long L, long long clip_stride, long long T, int hop, int center,
                       const float* window, const float2* tw, const float2* tw512, float2* out, float* phase,
                       hipStream_t stream) {
  const long long nframes = B * T;
  if (nframes == 0) return 0;
  P5 p = {};
  p.x = x; p.window = window; p.tw = tw; p.tw512 = tw512; p.X = out; p.phase_out = phase;
  p.L = L; p.clip_stride = clip_stride; p.T = T; p.total_frames = nframes; p.hop = hop; p.center = center;
  // the sliding-window / aligned-stream kernel: torch.stft's framing at hop n/4, 8-byte aligned clips, a 512-byte aligned
  // output, no phase side output
  if (center && hop == 128 && !phase && L >= 512 && (clip_stride & 1) == 0 && (((uintptr_t)x) & 7) == 0 &&
      (((uintptr_t)out) & 511) == 0 && (((uintptr_t)window) & 7) == 0 && variant(kVarFrameKernels) == 0) {
    P5Run q = {};
    q.x = x; q.window = window; q.tw = tw; q.tw512 = tw512; q.X = out;
    q.B = B; q.L = L; q.clip_stride = clip_stride; q.T = T;
    const long long slots = resident_waves(stft512_run_fwd_kernel, 64 * W5R, 0);
    const long long pairs = (T + 1) / 2;
    q.pairs_per_run = plan_units_per_run(B, pairs, slots, 8, 1);
    q.runs_per_clip = (pairs + q.pairs_per_run - 1) / q.pairs_per_run;
    const long long waves = B * q.runs_per_clip;
    hipLaunchKernelGGL(stft512_run_fwd_kernel, dim3((unsigned)((waves + W5R - 1) / W5R)), dim3(64 * W5R), 0, stream, q);
    return hipGetLastError() == hipSuccess ? 0 : -5;
  }
  const long long npairs = (nframes + 1) / 2;
  p.pairs_per_block = pairs_per_block_5(npairs);
  const long long blocks = (npairs + p.pairs_per_block - 1) / p.pairs_per_block;
  if (phase) hipLaunchKernelGGL(stft512_fwd_kernel<true>, dim3((unsigned)blocks), dim3(64 * W5), 0, stream, p);
  else hipLaunchKernelGGL(stft512_fwd_kernel<false>, dim3((unsigned)blocks), dim3(64 * W5), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft512_frames(const float2* X, const float* mag, const float* phase, long long nframes, const float* window,
                           const float2* tw, const float2* tw512, float* frames, hipStream_t stream) {
  if (nframes == 0) return 0;
  P5 p = {};
  p.X = const_cast<float2*>(X); p.mag = mag; p.phase = phase; p.window = window; p.tw = tw; p.tw512 = tw512; p.y = frames;
  p.total_frames = nframes;
  const long long npairs = (nframes + 1) / 2;
  p.pairs_per_block = pairs_per_block_5(npairs);
  const long long blocks = (npairs + p.pairs_per_block - 1) / p.pairs_per_block;
  if (X) hipLaunchKernelGGL(irfft512_frames_kernel<false>, dim3((unsigned)blocks), dim3(64 * W5), 0, stream, p);
  else hipLaunchKernelGGL(irfft512_frames_kernel<true>, dim3((unsigned)blocks), dim3(64 * W5), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
