// fft512.h -- one-wavefront 512-point complex FFT for gfx950 (wave64).
//
// A 1024-point real transform (n_fft = 1024, the reference's default:
// transforms/stft.py:34) is computed as a 512-point complex FFT of
// z[n] = x[2n] + i x[2n+1] plus a real split/merge pass.  512 = 8*8*8, so the
// complex FFT is three radix-8 passes with exactly 8 complex points per lane of
// a 64-lane wavefront.  Between passes the wave transposes its data through a
// private LDS slab (no s_barrier: LDS operations of one wave execute in order).
//
// Index algebra (forward, W_n = exp(-2*pi*i/n); inverse uses conjugates):
//   n = n0 + 8 n1 + 64 n2,  k = k0 + 8 k1 + 64 k2
//   pass 1: lane l=(n0,n1) holds z[l + 64 m], m=n2;  A[k0]  = sum_m  z W8^{m k0};  *= W512^{l k0}
//   xchg 1: -> lane (n0,k0) holds n1 = 0..7           (LDS index n1*72 + n0 + 8 k0)
//   pass 2:                                            B[k1]  = sum_n1 A W8^{n1 k1}; *= W64^{n0 k1}
//   xchg 2: -> lane (k0,k1) holds n0 = 0..7           (LDS index n0*66 + k0 + 8 k1)
//   pass 3:                                            Z[lane + 64 k2] = sum_n0 B W8^{n0 k2}
// Both LDS layouts are bank-conflict free for ds_write_b64 (16-lane groups hit
// 16 distinct 8-byte slots of a 128-byte row) and ds_read_b64 (32 consecutive
// lanes read 256 contiguous bytes).
#pragma once
#include <hip/hip_runtime.h>

namespace at_hip {

constexpr int kFftLdsFloat2PerWave = 568;  // 7*72 + 63 + 1

// twiddle table layout (float2 each), built on the host in double precision:
//   [0      .. 7*64)   tw1[k0-1][lane]  = W512^{lane*k0}          k0 = 1..7
//   [7*64   .. 14*64)  tw2[k1-1][lane]  = W64^{(lane&7)*k1}       k1 = 1..7
//   [14*64  .. 22*64)  twr[m][lane]     = W1024^{lane + 64 m}     m  = 0..7
constexpr int kTwiddleCount = 22 * 64;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 mul_w4(float2 a) {
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// multiply by W8 = (1 -/+ i)/sqrt2
template <bool INV>
__device__ __forceinline__ float2 mul_w8(float2 a) {
  const float r = 0.70710678118654752440f;
  return INV ? make_float2((a.x - a.y) * r, (a.x + a.y) * r) : make_float2((a.x + a.y) * r, (a.y - a.x) * r);
}
// multiply by W8^3 = (-1 -/+ i)/sqrt2
template <bool INV>
__device__ __forceinline__ float2 mul_w8_3(float2 a) {
  const float r = 0.70710678118654752440f;
  return INV ? make_float2((-a.x - a.y) * r, (a.x - a.y) * r) : make_float2((a.y - a.x) * r, (-a.x - a.y) * r);
}

// in-register 8-point DFT, natural-order output: v[k] = sum_n v[n] W8^{nk}
template <bool INV>
__device__ __forceinline__ void radix8(float2 (&v)[8]) {
  float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  float2 a1 = cadd(v[1], v[5]), a5 = mul_w8<INV>(csub(v[1], v[5]));
  float2 a2 = cadd(v[2], v[6]), a6 = mul_w4<INV>(csub(v[2], v[6]));
  float2 a3 = cadd(v[3], v[7]), a7 = mul_w8_3<INV>(csub(v[3], v[7]));
  float2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
  float2 b1 = cadd(a1, a3), b3 = mul_w4<INV>(csub(a1, a3));
  float2 b4 = cadd(a4, a6), b6 = csub(a4, a6);
  float2 b5 = cadd(a5, a7), b7 = mul_w4<INV>(csub(a5, a7));
  v[0] = cadd(b0, b1);
  v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3);
  v[6] = csub(b2, b3);
  v[1] = cadd(b4, b5);
  v[5] = csub(b4, b5);
  v[3] = cadd(b6, b7);
  v[7] = csub(b6, b7);
}

// compiler-level ordering of this wave's LDS traffic (the hardware already
// executes one wave's DS operations in order)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// twiddles held in registers (loop invariant, 44 VGPRs) ...
struct Twiddles {
  float2 t1[7];
  float2 t2[7];
  float2 tr[8];
  __device__ __forceinline__ float2 get1(int k) const { return t1[k]; }
  __device__ __forceinline__ float2 get2(int k) const { return t2[k]; }
  __device__ __forceinline__ float2 getr(int m) const { return tr[m]; }
};

// ... or read at the point of use from a workgroup-shared LDS copy of the table
// (tab[row * 64 + lane]: consecutive lanes, conflict-free ds_read_b64), trading 22 LDS reads per
// frame for 44 VGPRs -- one more resident wave per SIMD in the streaming kernels.
template <bool INV>
struct LdsTwiddles {
  const float2* tab;  // kTwiddleCount float2 in LDS, forward-sign values
  int lane;
  __device__ __forceinline__ float2 fix(float2 a) const { return INV ? make_float2(a.x, -a.y) : a; }
  __device__ __forceinline__ float2 get1(int k) const { return fix(tab[k * 64 + lane]); }
  __device__ __forceinline__ float2 get2(int k) const { return fix(tab[(7 + k) * 64 + lane]); }
  __device__ __forceinline__ float2 getr(int m) const { return fix(tab[(14 + m) * 64 + lane]); }
};

template <bool INV>
__device__ __forceinline__ void load_twiddles(Twiddles& tw, const float2* __restrict__ tab, int lane) {
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    float2 a = tab[k * 64 + lane];
    float2 b = tab[(7 + k) * 64 + lane];
    tw.t1[k] = INV ? cconj(a) : a;
    tw.t2[k] = INV ? cconj(b) : b;
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    float2 c = tab[(14 + m) * 64 + lane];
    tw.tr[m] = INV ? cconj(c) : c;
  }
}

// 512-point complex FFT of one wave.  In: v[m] = z[lane + 64 m].
// Out: v[m] = Z[lane + 64 m] (unnormalised).  `lds` is this wave's private slab
// of kFftLdsFloat2PerWave float2.
template <bool INV, typename TW>
__device__ __forceinline__ void fft512(float2 (&v)[8], const TW& tw, float2* lds, int lane) {
  const int lo = lane & 7, hi = lane >> 3;
  radix8<INV>(v);
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw.get1(k - 1));
  // xchg 1: writer (n0=lo, n1=hi), element k0 -> index n1*72 + n0 + 8*k0
#pragma unroll
  for (int k = 0; k < 8; ++k) lds[hi * 72 + lo + 8 * k] = v[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = lds[k * 72 + lane];
  wave_lds_sync();
  radix8<INV>(v);
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw.get2(k - 1));
  // xchg 2: writer (n0=lo, k0=hi), element k1 -> index n0*66 + k0 + 8*k1
#pragma unroll
  for (int k = 0; k < 8; ++k) lds[lo * 66 + hi + 8 * k] = v[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = lds[k * 66 + lane];
  wave_lds_sync();
  radix8<INV>(v);
}

// Fetch the mirror partner P[m] = Z[(512 - (lane + 64 m)) mod 512] of every element.
__device__ __forceinline__ void mirror512(const float2 (&v)[8], float2 (&p)[8], int lane) {
  const int src = (64 - lane) & 63;
  float2 q[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
  // lane 0: partner of k = 64 m is 64 (8 - m) -> own register (8-m)&7
  // lane>0: partner lives in lane 64-lane, register 7-m
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    float2 a = q[7 - m];
    float2 b = q[(8 - m) & 7];
    p[m] = (lane == 0) ? b : a;
  }
}

// real-FFT merge after the forward complex FFT:
//   X[k] = (Z[k] + conj Z[512-k])/2 - (i/2) W1024^k (Z[k] - conj Z[512-k]),  k = lane + 64 m
// returns X[512] (Nyquist) in `nyq` (meaningful on lane 0 only).
template <typename TW>
__device__ __forceinline__ void rfft_merge(float2 (&v)[8], const TW& tw, int lane, float2& nyq) {
  float2 p[8];
  mirror512(v, p, lane);
  nyq = make_float2(v[0].x - v[0].y, 0.0f);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    float2 zp = cconj(p[m]);
    float2 e = make_float2(0.5f * (v[m].x + zp.x), 0.5f * (v[m].y + zp.y));
    float2 d = make_float2(0.5f * (v[m].x - zp.x), 0.5f * (v[m].y - zp.y));
    float2 wd = cmul(tw.getr(m), d);  // W^k * d
    // -i * wd = (wd.y, -wd.x)
    v[m] = make_float2(e.x + wd.y, e.y - wd.x);
  }
}

// inverse of rfft_merge: from one-sided X (X[512] passed as xnyq_re to lane 0)
// build Z[k] = E[k] + i O[k] with E = (X[k] + conj X[512-k]), O = (X[k] - conj X[512-k]) conj(W1024^k)
// (the common factor 1/2 is folded into the caller's 1/N scale).
// `tw` must have been loaded with INV = true (tr = conj W1024^k).
template <typename TW>
__device__ __forceinline__ void irfft_split(float2 (&v)[8], const TW& tw, int lane, float xnyq_re) {
  // c2r ignores the imaginary parts of DC and Nyquist
  if (lane == 0) v[0].y = 0.0f;
  float2 p[8];
  mirror512(v, p, lane);
  if (lane == 0) p[0] = make_float2(xnyq_re, 0.0f);  // partner of k=0 is X[512]
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    float2 xp = cconj(p[m]);
    float2 e = cadd(v[m], xp);
    float2 d = cmul(csub(v[m], xp), tw.getr(m));
    // Z = e + i d
    v[m] = make_float2(e.x - d.y, e.y + d.x);
  }
}

}  // namespace at_hip
