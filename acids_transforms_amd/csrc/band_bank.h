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

}  // namespace at_hip
