// band_bank.h -- the banded filterbank walk shared by the fused STFT epilogue (stft1024.hip) and the stand-alone
// banded projection (mel_banded.hip).
#pragma once

namespace at_hip {

// Banded filterbank for the fused |X| -> mel epilogue (host side: utils/banded.py).  In pass q lane l sums
// filter lane_filter[q*64+l] (-1: none) over pass_len[q] bins starting at bin lane_start[q*64+l] (a multiple
// of 4), four bins per step.  The weights are stored pass-major, step-major, lane-minor -- the float4 of
// step j sits at float4 index (quad_base[q] + j) * 64 + l -- so a wave's weight read is 1 KB contiguous
// (conflict-free ds_read_b128); the host picks the lanes so that the magnitude reads do not collide either.
constexpr int kMaxBandPasses = 40;   // 64 filters per pass: banks of up to 2560 filters (the reference's default bank has
                                     // n_fft/2+1: 513 at n_fft 1024, 2049 at 4096); the fused epilogue takes at most 16
struct BandBank {
  const int* lane_filter;
  const int* lane_start;
  const float* weights;
  int n_filters, n_passes;
  int pass_len[kMaxBandPasses];   // bins walked in each pass (multiple of 4)
};
// where the fused STFT kernel puts the two halves of a Polar representation
struct PolarOut {
  float* phase;
  long long feat_ld, phase_ld;
  const float* ph_offset;
  const float* ph_scale;
};
constexpr int kMaxBandFloats = 8192;   // fused epilogue: LDS copy of the weights (dynamic LDS), 64 * sum(pass_len) floats <= 32 KB
constexpr int kMaxFusedPasses = 16;    // fused epilogue: passes (the stand-alone projection sizes itself by the LDS budget)

// The fixed-length walk of a two-pass bank (pass lengths FQ0 >= FQ1 quads, compile-time): this lane's two band sums
// from an LDS row of magnitudes.  Shared by the fused n_fft-1024 epilogue (stft1024.hip) and the stand-alone
// fixed-form projection (mel_banded.hip) so that the two give the same bits for the same spectrum.  The long pass runs
// on two independent packed sums (even / odd quads), joined at the end.
typedef float bb_v2f __attribute__((ext_vector_type(2)));
template <int FQ0, int FQ1>
__device__ __forceinline__ void band_walk_fixed(const float* absrow, int start0, int start1, const float* wlds, int lane,
                                                float& sum0, float& sum1) {
  const float4* a0 = reinterpret_cast<const float4*>(absrow + start0);
  const float4* a1 = reinterpret_cast<const float4*>(absrow + start1);
  const float4* w0 = reinterpret_cast<const float4*>(wlds) + lane;
  const float4* w1 = w0 + FQ0 * 64;
  bb_v2f s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < FQ1; ++j) {
    const float4 av = a1[j], wv = w1[j * 64];
    s1 = __builtin_elementwise_fma((bb_v2f){av.x, av.y}, (bb_v2f){wv.x, wv.y}, s1);
    s1 = __builtin_elementwise_fma((bb_v2f){av.z, av.w}, (bb_v2f){wv.z, wv.w}, s1);
  }
#pragma unroll
  for (int j = 0; j < FQ0; ++j) {
    const float4 av = a0[j], wv = w0[j * 64];
    if (j & 1) {
      s2 = __builtin_elementwise_fma((bb_v2f){av.x, av.y}, (bb_v2f){wv.x, wv.y}, s2);
      s2 = __builtin_elementwise_fma((bb_v2f){av.z, av.w}, (bb_v2f){wv.z, wv.w}, s2);
    } else {
      s0 = __builtin_elementwise_fma((bb_v2f){av.x, av.y}, (bb_v2f){wv.x, wv.y}, s0);
      s0 = __builtin_elementwise_fma((bb_v2f){av.z, av.w}, (bb_v2f){wv.z, wv.w}, s0);
    }
  }
  s0 += s2;
  sum0 = s0.x + s0.y;
  sum1 = s1.x + s1.y;
}

// The fixed-length walk of a bank with up to eight passes, their lengths in quads packed one per nibble of FQP (pass 0 in
// the low nibble) -- the reference's default bank, Magnitude() at sr 44100 / n_fft 1024: 404 non-empty filters of 513 in
// seven passes of 3, 2, 2, 1, 1, 1, 1 quads and two passes of empty filters (NP = 9, FQP = 0x001111223: a pass of zero quads
// sums nothing and its lanes emit contrast(0)).  a_off[q]: byte offset of this lane's walk of pass q inside the LDS row of
// magnitudes.  Every pass is straight-line code on its own packed sum.
typedef unsigned long long fqp_t;
constexpr int fqp_quads(fqp_t fqp, int q) { return (int)((fqp >> (4 * q)) & 15u); }
constexpr int fqp_base(fqp_t fqp, int q) {
  int b = 0;
  for (int i = 0; i < q; ++i) b += fqp_quads(fqp, i);
  return b;
}
template <fqp_t FQP, int NP>
__device__ __forceinline__ void band_walk_packed(const float* absrow, const int (&a_off)[NP], const float* wlds, int lane,
                                                 float (&sum)[NP]) {
  const float4* w = reinterpret_cast<const float4*>(wlds) + lane;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const float4* a = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(absrow) + a_off[q]);
    bb_v2f s = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < fqp_quads(FQP, q); ++j) {
      const float4 av = a[j], wv = w[(fqp_base(FQP, q) + j) * 64];
      s = __builtin_elementwise_fma((bb_v2f){av.x, av.y}, (bb_v2f){wv.x, wv.y}, s);
      s = __builtin_elementwise_fma((bb_v2f){av.z, av.w}, (bb_v2f){wv.z, wv.w}, s);
    }
    sum[q] = s.x + s.y;
    // Keep the scheduler from hoisting every pass's reads to the top (22 ds_read_b128 = 88 registers for the default
    // bank: 200 bytes of scratch per lane and 1.86 ms instead of 1.30): a fence whenever another three quads are done.
    if (fqp_base(FQP, q + 1) / 3 != fqp_base(FQP, q) / 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// Contrast of the fused / fixed-form epilogues.  The arguments are >= eps = 1.19e-7 (never denormal), so the hardware
// log2 (1 ulp) times ln 2 / log10 2 is within ~2 ulp of logf / log10f at a sixth of the instructions.
__device__ __forceinline__ float band_contrast_fast(float v, int mode, float eps) {
  switch (mode) {
    case 1: return __builtin_amdgcn_logf(1.0f + v) * 0.69314718055994530942f;
    case 2: return __builtin_amdgcn_logf(fmaxf(v, eps)) * 0.69314718055994530942f;
    case 3: return __builtin_amdgcn_logf(fmaxf(v, eps)) * 0.30102999566398119521f;
    default: return v;
  }
}

}  // namespace at_hip
