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

// s_setprio of the forward kernels' phases (stft1024.hip: transform > stores > epilogue); -DAT_WAVE_PRIORITY=0 builds
// without it for A/B runs
#ifndef AT_WAVE_PRIORITY
#define AT_WAVE_PRIORITY 1
#endif
template <int LEVEL>
__device__ __forceinline__ void wave_priority() {
#if AT_WAVE_PRIORITY
  __builtin_amdgcn_s_setprio(LEVEL);
#endif
}

constexpr int kFftLdsFloat2PerWave = 568;  // 7*72 + 63 + 1

// twiddle table layout (float2 each), built on the host in double precision:
//   [0      .. 7*64)   tw1[k0-1][lane]  = W512^{lane*k0}          k0 = 1..7
//   [7*64   .. 14*64)  tw2[k1-1][lane]  = W64^{(lane&7)*k1}       k1 = 1..7
//   [14*64  .. 22*64)  twr[m][lane]     = W1024^{lane + 64 m}     m  = 0..7
constexpr int kTwiddleCount = 22 * 64;
constexpr int kTileCtrSlots = 4096;   // {next tile, workgroups done} pairs behind the twiddle table (persistent kernels)

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// ---------------------------------------------------------------------------
// Complex arithmetic on packed fp32 (v_pk_*_f32: one instruction, both components).  A complex number is a
// 64-bit VGPR pair {re, im}.  The compiler folds operand *swaps* (op_sel) into packed instructions but not a
// negation of one half only, and without that a complex multiply costs five instructions instead of two and
// every multiplication by +-i an extra one (the auto-vectorised scalar version of this file: 372 VALU per
// 1024-point frame, 119 of them moves).  The handful of operations that need a per-half sign are therefore
// spelled as single instructions with explicit op_sel / neg modifiers:
//   op_sel[i]    : half of source i that feeds the LOW result   (0 = low half)
//   op_sel_hi[i] : half of source i that feeds the HIGH result  (1 = high half)
//   neg_lo / neg_hi[i] : negate source i in the low / high result
// ---------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f to_v(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ float2 to_f2(v2f a) { return make_float2(a.x, a.y); }

// a + (-i) b = {a.x + b.y, a.y - b.x}
__device__ __forceinline__ v2f add_mi(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a + (+i) b = {a.x - b.y, a.y + b.x}
__device__ __forceinline__ v2f add_pi(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a + r (-i) b = {a.x + r b.y, a.y - r b.x};  rr = {r, r} in scalar registers
__device__ __forceinline__ v2f fma_mi(v2f a, v2f b, v2f rr) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(b), "s"(rr), "v"(a));
  return r;
}
// a + r (+i) b = {a.x - r b.y, a.y + r b.x}
__device__ __forceinline__ v2f fma_pi(v2f a, v2f b, v2f rr) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(b), "s"(rr), "v"(a));
  return r;
}
// W4 = -i (forward) / +i (inverse):  a + W4 b,  a - W4 b,  a + r W4 b,  a - r W4 b
template <bool INV> __device__ __forceinline__ v2f rot_add(v2f a, v2f b) { return INV ? add_pi(a, b) : add_mi(a, b); }
template <bool INV> __device__ __forceinline__ v2f rot_sub(v2f a, v2f b) { return INV ? add_mi(a, b) : add_pi(a, b); }
template <bool INV> __device__ __forceinline__ v2f rot_fma(v2f a, v2f b, v2f rr) { return INV ? fma_pi(a, b, rr) : fma_mi(a, b, rr); }
template <bool INV> __device__ __forceinline__ v2f rot_fms(v2f a, v2f b, v2f rr) { return INV ? fma_mi(a, b, rr) : fma_pi(a, b, rr); }

// a * w = {a.x w.x - a.y w.y, a.x w.y + a.y w.x}: two instructions
__device__ __forceinline__ v2f cmul_v(v2f a, v2f w) {
  const v2f t = a.yy * w.yx;   // {a.y w.y, a.y w.x}: a plain packed multiply with operand swaps
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// a * conj(w) = {a.x w.x + a.y w.y, a.y w.x - a.x w.y}: the inverse transform uses the forward table as is
__device__ __forceinline__ v2f cmul_conj_v(v2f a, v2f w) {
  const v2f t = a.yy * w.yx;
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
template <bool INV> __device__ __forceinline__ v2f twiddle(v2f a, v2f w) { return INV ? cmul_conj_v(a, w) : cmul_v(a, w); }
// a + conj(b), a - conj(b)
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// h a + (-i) b = {h a.x + b.y, h a.y - b.x};  hh = {h, h} in scalar registers
__device__ __forceinline__ v2f scale_add_mi(v2f a, v2f hh, v2f b) {
  v2f r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "s"(hh), "v"(b));
  return r;
}

// in-register 8-point DFT, natural-order output: v[k] = sum_n v[n] W8^{nk}.  26 packed instructions:
// W8 = (1 + W4)/sqrt2 and W8^3 = (W4 - 1)/sqrt2, so the two odd rotations are one rotate-add each, their common
// factor 1/sqrt2 rides on the last level's multiply-adds, and every multiplication by W4 is an operand swap.
template <bool INV>
__device__ __forceinline__ void radix8(v2f (&v)[8]) {
  const v2f rr = {0.70710678118654752440f, 0.70710678118654752440f};
  const v2f a0 = v[0] + v[4], a4 = v[0] - v[4];
  const v2f a1 = v[1] + v[5], d1 = v[1] - v[5];
  const v2f a2 = v[2] + v[6], d2 = v[2] - v[6];
  const v2f a3 = v[3] + v[7], d3 = v[3] - v[7];
  const v2f s5 = rot_add<INV>(d1, d1);        // sqrt2 W8 d1
  const v2f s7 = rot_sub<INV>(d3, d3);        // -sqrt2 W8^3 d3
  const v2f b0 = a0 + a2, b2 = a0 - a2;
  const v2f b1 = a1 + a3, d13 = a1 - a3;
  const v2f b4 = rot_add<INV>(a4, d2), b6 = rot_sub<INV>(a4, d2);
  const v2f u5 = s5 - s7, u7 = s5 + s7;       // sqrt2 (a5 + a7), sqrt2 (a5 - a7)
  v[0] = b0 + b1;
  v[4] = b0 - b1;
  v[2] = rot_add<INV>(b2, d13);
  v[6] = rot_sub<INV>(b2, d13);
  v[1] = __builtin_elementwise_fma(u5, rr, b4);
  v[5] = __builtin_elementwise_fma(-u5, rr, b4);
  v[3] = rot_fma<INV>(b6, u7, rr);
  v[7] = rot_fms<INV>(b6, u7, rr);
}

// One 8-byte LDS read that stays one `ds_read_b64`.  Left to itself the compiler pairs neighbouring 8-byte reads
// into `ds_read2_b64` / `ds_read2st64_b64`, which the LDS serves at HALF the rate (MI355X_MICROARCH.md, LDS
// table: ds_read_b64 2 cycles per wave-instruction, 256 B/clk/CU; ds_read2_b64 8 cycles for twice the bytes,
// 128 B/clk/CU).  These kernels keep the LDS array busier than any other unit (exchanges, twiddles, window:
// 46 reads per frame in the fused forward), so the pairing cost a quarter of their LDS time.  A volatile access is
// not merged; the compiler still schedules arithmetic around it and keeps its lgkmcnt bookkeeping.
#ifndef AT_LDS_NOMERGE
#define AT_LDS_NOMERGE 1
#endif
template <typename T>
__device__ __forceinline__ T lds_read_single(const T* p) {
#if AT_LDS_NOMERGE
  // the pointer is known to be LDS: say so, or the volatile access is lowered as a flat load
  typedef const volatile __attribute__((address_space(3))) T* lds_ptr;
  return *(lds_ptr)(p);
#else
  return *p;
#endif
}

// compiler-level ordering of this wave's LDS traffic (the hardware already
// executes one wave's DS operations in order)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Twiddles are stored with the forward sign in every kernel (the inverse multiplies by the conjugate in the
// same two instructions).  `tr` holds W1024^k / 2 in forward kernels (the real-FFT merge wants the half) and
// W1024^k in inverse ones: `TR_SCALE`.
// twiddles held in registers (loop invariant, 44 VGPRs) ...
struct Twiddles {
  v2f t1[7];
  v2f t2[7];
  v2f tr[8];
  __device__ __forceinline__ v2f get1(int k) const { return t1[k]; }
  __device__ __forceinline__ v2f get2(int k) const { return t2[k]; }
  __device__ __forceinline__ v2f getr(int m) const { return tr[m]; }
};

// ... or read at the point of use from a workgroup-shared LDS copy of the table
// (tab[row * 64 + lane]: consecutive lanes, conflict-free ds_read_b64), trading 22 LDS reads per
// frame for 44 VGPRs -- one more resident wave per SIMD in the streaming kernels.  The kernel that fills
// the LDS copy applies the forward 1/2 to rows 14..21 (`twiddle_for_lds`).
template <bool INV>
struct LdsTwiddles {
  const float2* tab;  // kTwiddleCount float2 in LDS
  int lane;
  __device__ __forceinline__ v2f get1(int k) const { return lds_read_single(reinterpret_cast<const v2f*>(tab) + k * 64 + lane); }
  __device__ __forceinline__ v2f get2(int k) const { return lds_read_single(reinterpret_cast<const v2f*>(tab) + (7 + k) * 64 + lane); }
  __device__ __forceinline__ v2f getr(int m) const { return lds_read_single(reinterpret_cast<const v2f*>(tab) + (14 + m) * 64 + lane); }
};
// ... or both: the two pass tables (14 rows, indexed by the physical lane) in registers, the W1024 rows of the real
// merge read from LDS through `col` -- what the kernels with rotated output columns need (the merge's twiddle follows
// the column, which changes from frame to frame), at 14 fewer LDS reads per frame than LdsTwiddles.
struct HybridTwiddles {
  v2f t1[7];
  v2f t2[7];
  const float2* tab;
  int col;
  __device__ __forceinline__ v2f get1(int k) const { return t1[k]; }
  __device__ __forceinline__ v2f get2(int k) const { return t2[k]; }
  __device__ __forceinline__ v2f getr(int m) const { return lds_read_single(reinterpret_cast<const v2f*>(tab) + (14 + m) * 64 + col); }
};

template <bool INV>
__device__ __forceinline__ float2 twiddle_for_lds(const float2* __restrict__ tab, int i) {
  const float2 a = tab[i];
  return (!INV && i >= 14 * 64) ? make_float2(0.5f * a.x, 0.5f * a.y) : a;
}

template <bool INV>
__device__ __forceinline__ void load_twiddles(Twiddles& tw, const float2* __restrict__ tab, int lane) {
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    tw.t1[k] = to_v(tab[k * 64 + lane]);
    tw.t2[k] = to_v(tab[(7 + k) * 64 + lane]);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) tw.tr[m] = to_v(twiddle_for_lds<INV>(tab, (14 + m) * 64 + lane));
}

// 512-point complex FFT of one wave.  In: v[m] = z[lane + 64 m].
// Out: v[m] = Z[lane + 64 m] (unnormalised).  `lds` is this wave's private slab
// of kFftLdsFloat2PerWave float2.
// `out_lane` (default: the lane itself): which lane's outputs this lane computes in the last pass, i.e. the lane
// ends up with Z[out_lane + 64 m].  The second exchange reads through it, so any permutation of the 64 output
// columns over the lanes is free (a rotation keeps the reads conflict-free: 32 consecutive lanes still cover 32
// consecutive 8-byte slots modulo a multiple of 256 bytes).
// AT_XCHG1_SWAP: 1 (default) = the forward transforms take their first exchange through registers (same-box A/B, 1024
// clips: plain forward 0.762 -> 0.758 ms, fused 0.884 -> 0.876, features only 0.616 -> 0.602; the inverse 0.722 -> 0.726,
// so it keeps the LDS exchange); 0 = LDS for both; 2 = registers for both
#ifndef AT_XCHG1_SWAP
#define AT_XCHG1_SWAP 1
#endif
__device__ __forceinline__ void xchg1_registers(v2f (&v)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k)             // register bit 2 <-> lane bit 5
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k][c]), __float_as_uint(v[k + 4][c]), false, false);
      v[k][c] = __uint_as_float(sw[0]);
      v[k + 4][c] = __uint_as_float(sw[1]);
    }
#pragma unroll
  for (int g = 0; g < 8; g += 4)          // register bit 1 <-> lane bit 4
#pragma unroll
    for (int k = g; k < g + 2; ++k)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[k][c]), __float_as_uint(v[k + 2][c]), false, false);
        v[k][c] = __uint_as_float(sw[0]);
        v[k + 2][c] = __uint_as_float(sw[1]);
      }
#pragma unroll
  for (int k = 0; k < 8; k += 2)          // register bit 0 <-> lane bit 3: row_ror:8 swaps the halves of a 16-lane row
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int a = __float_as_int(v[k][c]), b = __float_as_int(v[k + 1][c]);
      const int a2 = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false);     // lanes 8..15 of each row take b[lane ^ 8]
      const int b2 = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);     // lanes 0..7 take a[lane ^ 8]
      v[k][c] = __int_as_float(a2);
      v[k + 1][c] = __int_as_float(b2);
    }
}

template <bool INV, typename TW>
__device__ __forceinline__ void fft512(v2f (&v)[8], const TW& tw, float2* lds_f2, int lane, int out_lane = -1) {
  if (out_lane < 0) out_lane = lane;
  v2f* lds = reinterpret_cast<v2f*>(lds_f2);
  const int lo = lane & 7, hi = lane >> 3;
  radix8<INV>(v);
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = twiddle<INV>(v[k], tw.get1(k - 1));
  if constexpr (AT_XCHG1_SWAP == 2 || (AT_XCHG1_SWAP == 1 && !INV)) {
    // xchg 1 in registers: the transpose (register index) <-> (lane bits 3..5) as three bit swaps -- v_permlane32_swap
    // (lane bit 5 with register bit 2), v_permlane16_swap (lane bit 4, register bit 1), and two masked DPP row rotations
    // by 8 (lane bit 3, register bit 0) -- 32 instructions instead of 4 ds_write2_b64 + 8 ds_read_b64 + two waits
    xchg1_registers(v);
  } else {
    // xchg 1: writer (n0=lo, n1=hi), element k0 -> index n1*72 + n0 + 8*k0
#pragma unroll
    for (int k = 0; k < 8; ++k) lds[hi * 72 + lo + 8 * k] = v[k];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lds_read_single(lds + k * 72 + lane);
    wave_lds_sync();
  }
  radix8<INV>(v);
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = twiddle<INV>(v[k], tw.get2(k - 1));
  // xchg 2: writer (n0=lo, k0=hi), element k1 -> index n0*66 + k0 + 8*k1
#pragma unroll
  for (int k = 0; k < 8; ++k) lds[lo * 66 + hi + 8 * k] = v[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = lds_read_single(lds + k * 66 + out_lane);
  wave_lds_sync();
  radix8<INV>(v);
}
// The fixed-length mel epilogues (stft1024.hip, FQ path) lay |X| over this slab as 513 floats and let a walk run on to
// float 639 over zero weights WITHOUT clearing floats 513..639 first: correct only while those floats -- float2 slots
// 256..319 -- hold finite leftovers of the SAME frame, i.e. while the two exchanges above rewrite every one of them in
// every transform (0 x Inf would be NaN, stale values of an earlier frame might be anything).  Pinned here, next to the
// strides it depends on.  With the first exchange in registers (AT_XCHG1_SWAP, the forward default since round
// 4) only the second one rewrites the slab and slots 262-263 are never reached: the kernels with a fixed-length epilogue
// clear floats 512..639 of the slab once, at their start (stft1024.hip), and everything written there afterwards is finite.
constexpr bool fft512_exchanges_cover(int lo_slot, int hi_slot) {
  for (int s = lo_slot; s <= hi_slot; ++s) {
    bool hit = false;
    for (int a = 0; a < 8 && !hit; ++a) {
      if (s >= a * 72 && s < a * 72 + 64) hit = true;     // xchg 1: hi * 72 + lo + 8 k
      if (s >= a * 66 && s < a * 66 + 64) hit = true;     // xchg 2: lo * 66 + hi + 8 k
    }
    if (!hit) return false;
  }
  return true;
}
static_assert(AT_XCHG1_SWAP || (fft512_exchanges_cover(256, 319) && 320 <= kFftLdsFloat2PerWave),
              "the fixed-length mel epilogue reads absrow[513..639] uncleared: both exchanges must rewrite float2 slots 256..319");

template <bool INV, typename TW>
__device__ __forceinline__ void fft512(float2 (&f)[8], const TW& tw, float2* lds, int lane) {
  v2f v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = to_v(f[m]);
  fft512<INV>(v, tw, lds, lane);
#pragma unroll
  for (int m = 0; m < 8; ++m) f[m] = to_f2(v[m]);
}

// Fetch the mirror partner P[m] = Z[(512 - (lane + 64 m)) mod 512] of every element.
__device__ __forceinline__ void mirror512(const v2f (&v)[8], v2f (&p)[8], int lane) {
  const int src = (64 - lane) & 63;
  v2f q[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
  // lane 0: partner of k = 64 m is 64 (8 - m) -> own register (8-m)&7
  // lane>0: partner lives in lane 64-lane, register 7-m
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const v2f a = q[7 - m];
    const v2f b = q[(8 - m) & 7];
    p[m] = (lane == 0) ? b : a;
  }
}

// The same with the output columns rotated over the lanes: this lane holds Z[col + 64 m], col = (lane - rot) & 63,
// and column c sits in lane (c + rot) & 63 -- the partner column (64 - col) & 63 in lane (2 rot - lane) & 63.
__device__ __forceinline__ void mirror512_rot(const v2f (&v)[8], v2f (&p)[8], int lane, int rot, int col) {
  const int src = (2 * rot - lane) & 63;
  v2f q[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    q[m].x = __shfl(v[m].x, src, 64);
    q[m].y = __shfl(v[m].y, src, 64);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const v2f a = q[7 - m];
    const v2f b = q[(8 - m) & 7];
    p[m] = (col == 0) ? b : a;
  }
}

// real-FFT merge after the forward complex FFT:
//   X[k] = (Z[k] + conj Z[512-k])/2 - (i/2) W1024^k (Z[k] - conj Z[512-k]),  k = lane + 64 m
// returns X[512] (Nyquist) in `nyq` (meaningful on lane 0 only).  `tw.getr` = W1024^k / 2 (forward tables).
template <typename TW>
__device__ __forceinline__ void rfft_merge(v2f (&v)[8], const TW& tw, int lane, float2& nyq) {
  const v2f hh = {0.5f, 0.5f};
  v2f p[8];
  mirror512(v, p, lane);
  nyq = make_float2(v[0].x - v[0].y, 0.0f);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const v2f e = add_conj(v[m], p[m]);              // Z + conj Z'
    const v2f d = sub_conj(v[m], p[m]);              // Z - conj Z'
    const v2f wd = cmul_v(d, tw.getr(m));            // (W/2) d
    v[m] = scale_add_mi(e, hh, wd);                  // e/2 - i wd
  }
}
// rfft_merge for rotated output columns (see mirror512_rot): `tw` must index its W1024 rows by the COLUMN (an
// LdsTwiddles built with lane = col); X[512] comes out on the lane whose column is 0.
template <typename TW>
__device__ __forceinline__ void rfft_merge_rot(v2f (&v)[8], const TW& tw, int lane, int rot, int col, float2& nyq) {
  const v2f hh = {0.5f, 0.5f};
  v2f p[8];
  mirror512_rot(v, p, lane, rot, col);
  nyq = make_float2(v[0].x - v[0].y, 0.0f);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const v2f e = add_conj(v[m], p[m]);
    const v2f d = sub_conj(v[m], p[m]);
    const v2f wd = cmul_v(d, tw.getr(m));
    v[m] = scale_add_mi(e, hh, wd);
  }
}
template <typename TW>
__device__ __forceinline__ void rfft_merge(float2 (&f)[8], const TW& tw, int lane, float2& nyq) {
  v2f v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = to_v(f[m]);
  rfft_merge(v, tw, lane, nyq);
#pragma unroll
  for (int m = 0; m < 8; ++m) f[m] = to_f2(v[m]);
}

// inverse of rfft_merge: from one-sided X (X[512] passed as xnyq_re to lane 0)
// build Z[k] = E[k] + i O[k] with E = (X[k] + conj X[512-k]), O = (X[k] - conj X[512-k]) conj(W1024^k)
// (the common factor 1/2 is folded into the caller's 1/N scale).  `tw.getr` = W1024^k (inverse tables).
template <typename TW>
__device__ __forceinline__ void irfft_split(v2f (&v)[8], const TW& tw, int lane, float xnyq_re) {
  // c2r ignores the imaginary parts of DC and Nyquist
  if (lane == 0) v[0].y = 0.0f;
  v2f p[8];
  mirror512(v, p, lane);
  if (lane == 0) p[0] = (v2f){xnyq_re, 0.0f};  // partner of k=0 is X[512]
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const v2f e = add_conj(v[m], p[m]);
    const v2f d = cmul_conj_v(sub_conj(v[m], p[m]), tw.getr(m));
    v[m] = add_pi(e, d);                              // Z = e + i d
  }
}
template <typename TW>
__device__ __forceinline__ void irfft_split(float2 (&f)[8], const TW& tw, int lane, float xnyq_re) {
  v2f v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = to_v(f[m]);
  irfft_split(v, tw, lane, xnyq_re);
#pragma unroll
  for (int m = 0; m < 8; ++m) f[m] = to_f2(v[m]);
}

}  // namespace at_hip
