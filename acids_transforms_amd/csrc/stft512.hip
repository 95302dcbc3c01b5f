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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
          const float2 e = env2[(size_t)mask * (HOP / 2) + u + 32 * j];
          dst[u + 32 * j] = make_float2(sum.x / e.x, sum.y / e.y);
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

int launch_stft512_fwd(const float* x, long long B, long long L, long long clip_stride, long long T, int hop, int center,
                       const float* window, const float2* tw, const float2* tw512, float2* out, float* phase,
                       hipStream_t stream) {
  const long long nframes = B * T;
  if (nframes == 0) return 0;
  P5 p = {};
  p.x = x; p.window = window; p.tw = tw; p.tw512 = tw512; p.X = out; p.phase_out = phase;
  p.L = L; p.clip_stride = clip_stride; p.T = T; p.total_frames = nframes; p.hop = hop; p.center = center;
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
