// mel_gemm.h -- parameter block of the exact-fp32 MFMA projection kernel (mel.hip), shared with sinebank.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace at_hip {

enum { A_COMPLEX_ABS = 0, A_COMPLEX_ABS2 = 1, A_REAL = 2, A_REAL_ABS = 3 };
enum { C_NONE = 0, C_LOG1P = 1, C_LOG = 2, C_LOG10 = 3 };

// contrast of the banded projections and its inverse (spectral_repr.py:190-208): shared by mel_banded.hip and the one-pass
// PolarIF.forward of phase_repr.hip, which must give the same bits.  phase_repr.hip is built with -ffp-contract=off, and the
// backend's expansion of the logarithm ends in a multiply-add that is fused or not with the CALL's contraction flag (one
// ulp apart): the builtins, not the header's logf (whose flag is the translation unit's), under a pragma that fixes it.
__device__ __forceinline__ float banded_contrast_fwd(float v, int mode, float eps) {
#pragma clang fp contract(fast)
  switch (mode) {
    case C_LOG1P: return __builtin_logf(1.0f + v);
    case C_LOG: return __builtin_logf(fmaxf(v, eps));
    case C_LOG10: return __builtin_log10f(fmaxf(v, eps));
    default: return v;
  }
}
__device__ __forceinline__ float banded_contrast_inv(float v, int mode, float eps) {
  switch (mode) {
    case C_LOG1P: return expf(v) - 1.0f;
    case C_LOG: return expf(v) - eps;
    case C_LOG10: return powf(10.0f, v);
    default: return v;
  }
}

struct MelParams {
  const void* A;       // rows x K  (complex64 or float32), row stride lda elements
  const float* Bm;     // K x N row-major, row stride ldb
  float* out;
  const float* offset; // device scalars (may be null => no normalisation)
  const float* scale;
  long long rows, lda, ld_out;
  long long T;         // >0: channel-major store out[(r/T)*N*T + n*T + r%T]  (MFCC layout)
  int K, N, ldb;
  int a_kind, contrast, inverse;  // inverse: prologue (x*scale+offset, invert_contrast) on A instead of epilogue
  float eps;
  int rs;              // LDS row stride in floats
  long long tiles_per_block;
  // optional: element offset added to A for each 128-column block (blockIdx.y) -- the sinebank contraction
  // reads a different frame of the (B, T, F) spectrum for every block of output samples
  const long long* a_block_offset;
  int dense;           // 1: skip the zero-block bookkeeping (bank known to be dense)
};


// out = epilogue(prologue(A) @ Bm), any K in [16, 576]; returns an AT_* code (mel.hip)
int launch_mel_project(const MelParams& p, hipStream_t stream);

}  // namespace at_hip
