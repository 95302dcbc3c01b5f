// mel.hip -- |X| -> filterbank contraction with fused contrast / normalise epilogue.
//
// Replaces (reference transforms/spectral_repr.py, transforms/mel.py):
//   x.abs(); torch.matmul(mag, mel_bank); contrast; Normalize.forward      :215-226   (K8,K9,K10)
//   Normalize.invert; invert_contrast; torch.matmul(mag, inverse_mel_bank) :228-240   (K10,K9,K11)
//   MelSpectrogram = |stft|^2 @ fbank, channel-major output                 mel.py:43-44,68-73 (K12)
//
// The contraction is a dense [rows x K] . [K x N] GEMM in exact fp32 on the
// matrix cores: v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fma chain, so the
// 1e-5 parity bound holds without any split-precision trick).
//
// Layout of one 512-thread workgroup (8 waves, one per CU):
//   * wave w owns 16 output columns of a 128-column block; its slice of the
//     bank (K x 16, <= 128 VGPRs) stays in registers for the whole launch;
//   * the workgroup walks 32-row tiles of the input; |.| is taken once while
//     the tile is staged in LDS (row stride 32q+2 floats: conflict-free
//     ds_read2_b64 A-fragments), double buffered, the next tile's global loads
//     interleaved with the current tile's MFMAs; one barrier per tile;
//   * epilogue straight from the accumulators (contrast, (x-offset)/scale).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acids_hip.h"
#include "mel_gemm.h"

namespace at_hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 32;
constexpr int THREADS = 512;

__device__ __forceinline__ float contrast_fwd(float v, int mode, float eps) {
  switch (mode) {
    case C_LOG1P: return logf(1.0f + v);
    case C_LOG: return logf(fmaxf(v, eps));
    case C_LOG10: return log10f(fmaxf(v, eps));
    default: return v;
  }
}

__device__ __forceinline__ float contrast_inv(float v, int mode, float eps) {
  switch (mode) {
    case C_LOG1P: return expf(v) - 1.0f;
    case C_LOG: return expf(v) - eps;
    case C_LOG10: return powf(10.0f, v);
    default: return v;
  }
}

// raw element of A as loaded from HBM (complex pair, or a real value in .x)
__device__ __forceinline__ float2 load_raw(const MelParams& p, long long row, int k, int colblock) {
  const long long at = row * p.lda + k + (p.a_block_offset ? p.a_block_offset[colblock] : 0);
  if (p.a_kind >= A_REAL) return make_float2(reinterpret_cast<const float*>(p.A)[at], 0.f);
  return reinterpret_cast<const float2*>(p.A)[at];
}

// what the contraction consumes: |.|, |.|^2, or the (de-normalised, de-contrasted) real value
__device__ __forceinline__ float finish_a(const MelParams& p, float2 c, float off, float sc) {
  if (p.a_kind >= A_REAL) {
    float v = c.x;
    if (p.a_kind == A_REAL_ABS) v = fabsf(v);
    if (p.inverse) {
      if (p.offset) v = __fadd_rn(__fmul_rn(v, sc), off);
      v = contrast_inv(v, p.contrast, p.eps);
    }
    return v;
  }
  const float s2 = fmaf(c.x, c.x, c.y * c.y);
  return (p.a_kind == A_COMPLEX_ABS2) ? s2 : __builtin_amdgcn_sqrtf(s2);  // |x| (overflow-safe hypot is not needed at audio scale)
}

__device__ __forceinline__ float load_a(const MelParams& p, long long row, int k, int colblock, float off, float sc) {
  return finish_a(p, load_raw(p, row, k, colblock), off, sc);
}

// stage rows [4c, 4c+4) x all K of tile `tile` (piece c): issue the global loads only
template <int NL>
__device__ __forceinline__ void piece_load(const MelParams& p, long long tile, int piece, float2 (&regs)[NL]) {
  const int tr = piece * 4 + (threadIdx.x >> 7);
  const long long row = tile * ROWS + tr;
  const int kseg = threadIdx.x & 127;
  const bool row_ok = row < p.rows;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int k = kseg + 128 * i;
    // all but the last 128-column segment are always inside K (NL = ceil(K/128))
    const bool ok = row_ok && (i + 1 < NL || k < p.K);
    regs[i] = ok ? load_raw(p, row, k, blockIdx.y) : make_float2(0.f, 0.f);
  }
}

// ... and, one MFMA group later, take |.| and park the piece in LDS
template <int NL>
__device__ __forceinline__ void piece_store(const MelParams& p, int piece, const float2 (&regs)[NL], float* buf,
                                            float off, float sc, int* nonfinite_flag) {
  const int tr = piece * 4 + (threadIdx.x >> 7);
  const int kseg = threadIdx.x & 127;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int k = kseg + 128 * i;
    if (i + 1 < NL || k < p.K) {
      const float v = finish_a(p, regs[i], off, sc);
      bad |= !(fabsf(v) <= 3.402823466e+38f);   // inf or NaN
      buf[tr * p.rs + k] = v;
    }
  }
  if (bad) *nonfinite_flag = 1;  // this tile must take the dense path (0 * NaN has to stay NaN)
}

// KSTEPS = number of 4-deep MFMA steps (K_main = 4*KSTEPS <= K); NL = ceil(K/128) loads per thread per piece
// DENSE: the bank has no all-zero blocks worth testing for (sinebank's oscillator matrix): no per-step mask
// test, so the MFMA chain of a staged piece is straight-line code and its LDS reads can run ahead
template <int KSTEPS, int NL, bool DENSE>
__global__ __launch_bounds__(THREADS) void mel_gemm_kernel(MelParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* buf0 = smem;
  float* buf1 = smem + ROWS * p.rs;
  int* flags = reinterpret_cast<int*>(smem + 2 * ROWS * p.rs);   // 3 rotating "tile holds inf/NaN" flags
  if (threadIdx.x < 3) flags[threadIdx.x] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int col = blockIdx.y * 128 + wave * 16 + li;
  const bool col_ok = col < p.N;
  const int kmain = 4 * KSTEPS;

  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }

  // this wave's slice of the bank: step s = 2j+e multiplies k = 8j + 2kk + e
  float breg[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int k = 8 * (s >> 1) + 2 * kk + (s & 1);
    breg[s] = col_ok ? p.Bm[(long long)k * p.ldb + col] : 0.0f;
  }

  const long long ntiles = (p.rows + ROWS - 1) / ROWS;
  long long tile = (long long)blockIdx.x * p.tiles_per_block;
  long long tile_end = tile + p.tiles_per_block;
  if (tile_end > ntiles) tile_end = ntiles;
  if (tile >= tile_end) return;

  // Zero-block skipping: a j-step (8 bank rows x this wave's 16 columns) whose bank entries are all
  // exactly zero contributes exactly 0 for finite inputs and is skipped (wave-uniform branch).  Mel
  // banks are banded, so most steps drop out; a dense bank skips nothing.  Tiles that contain an
  // inf/NaN take the dense path so that non-finite values propagate as in a dense matmul.
  unsigned long long nzmask = 0ull;
#pragma unroll
  for (int j = 0; j < KSTEPS / 2; ++j)
    if (__ballot(breg[2 * j] != 0.0f || breg[2 * j + 1] != 0.0f) != 0ull) nzmask |= 1ull << j;
  nzmask = __builtin_amdgcn_readfirstlane((unsigned)nzmask) |
           ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(nzmask >> 32)) << 32);

  // K tail (k >= 4*KSTEPS, at most 4 columns kept in registers; longer tails re-read the bank)
  float btail[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = kmain + q;
    btail[q] = (col_ok && k < p.K) ? p.Bm[(long long)k * p.ldb + col] : 0.0f;
  }

  // prologue: stage the first tile
  {
    float2 regs[NL];
    for (int c = 0; c < 8; ++c) {
      piece_load<NL>(p, tile, c, regs);
      piece_store<NL>(p, c, regs, buf0, off, sc, &flags[0]);
    }
  }
  __syncthreads();

  float* cur = buf0;
  float* nxt = buf1;
  for (; tile < tile_end; ++tile) {
    const bool has_next = tile + 1 < tile_end;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const long long it = tile - (long long)blockIdx.x * p.tiles_per_block;
    int* flag_cur = flags + (int)(it % 3);
    int* flag_nxt = flags + (int)((it + 1) % 3);
    if (threadIdx.x == 0) flags[(int)((it + 2) % 3)] = 0;   // next written one barrier from now
    const unsigned long long jmask = (__builtin_amdgcn_readfirstlane(*flag_cur) != 0) ? ~0ull : nzmask;
    const float* a0p = cur + li * p.rs + 2 * kk;
    const float* a1p = cur + (16 + li) * p.rs + 2 * kk;
    constexpr int CH = (KSTEPS / 2 + 7) / 8;  // j-iterations per staged piece
    // two pieces of the next tile in flight: piece c+1 is requested before the MFMAs of group c,
    // piece c is converted and parked after them
    float2 regs[2][NL];
    if (has_next) piece_load<NL>(p, tile + 1, 0, regs[0]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (has_next && c + 1 < 8) piece_load<NL>(p, tile + 1, c + 1, regs[(c + 1) & 1]);
#pragma unroll
      for (int jj = 0; jj < CH; ++jj) {
        const int j = c * CH + jj;
        if (j < KSTEPS / 2 && (DENSE || ((jmask >> j) & 1ull))) {
          const float2 a0 = *reinterpret_cast<const float2*>(a0p + 8 * j);
          const float2 a1 = *reinterpret_cast<const float2*>(a1p + 8 * j);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, breg[2 * j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, breg[2 * j], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, breg[2 * j + 1], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, breg[2 * j + 1], acc1, 0, 0, 0);
        }
      }
      if (has_next) piece_store<NL>(p, c, regs[c & 1], nxt, off, sc, flag_nxt);
    }
    // K tail (k >= 4*KSTEPS) on the vector ALU; C/D layout: col = lane&15, row = 4*(lane>>4) + reg
    for (int k = kmain; k < p.K; ++k) {
      const int q = k - kmain;
      const float b = (q < 4) ? btail[q & 3] : (col_ok ? p.Bm[(long long)k * p.ldb + col] : 0.0f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0[r] = fmaf(cur[(4 * kk + r) * p.rs + k], b, acc0[r]);
        acc1[r] = fmaf(cur[(16 + 4 * kk + r) * p.rs + k], b, acc1[r]);
      }
    }
    // epilogue
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long row = tile * ROWS + h * 16 + 4 * kk + r;
        float v = h ? acc1[r] : acc0[r];
        if (!p.inverse) {
          v = contrast_fwd(v, p.contrast, p.eps);
          if (p.offset) v = (v - off) / sc;
        }
        if (row < p.rows && col_ok) {
          if (p.T > 0) {
            const long long b = row / p.T, t = row - b * p.T;
            p.out[(b * p.N + col) * p.T + t] = v;
          } else {
            p.out[row * p.ld_out + col] = v;
          }
        }
      }
    }
    __syncthreads();
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
}

// fallback for shapes the MFMA kernel does not cover (K > 576): one thread per output
__global__ void mel_gemm_simple_kernel(MelParams p) {
  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }
  const long long total = p.rows * p.N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / p.N;
    const int col = (int)(i - row * p.N);
    float acc = 0.f;
    for (int k = 0; k < p.K; ++k) acc = fmaf(load_a(p, row, k, col >> 7, off, sc), p.Bm[(long long)k * p.ldb + col], acc);
    if (!p.inverse) {
      acc = contrast_fwd(acc, p.contrast, p.eps);
      if (p.offset) acc = (acc - off) / sc;
    }
    if (p.T > 0) {
      const long long b = row / p.T, t = row - b * p.T;
      p.out[(b * p.N + col) * p.T + t] = acc;
    } else {
      p.out[row * p.ld_out + col] = acc;
    }
  }
}

// no projection: |x| (or real x) -> contrast -> normalise, or the inverse chain  (Magnitude with mel=False)
struct PointParams {
  const void* A;
  float* out;
  const float* offset;
  const float* scale;
  long long n;
  int a_kind, contrast, inverse;
  float eps;
};

__global__ void mag_pointwise_kernel(PointParams p) {
  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x) {
    float v;
    if (p.a_kind >= A_REAL) {
      v = reinterpret_cast<const float*>(p.A)[i];
      if (p.a_kind == A_REAL_ABS) v = fabsf(v);
    } else {
      float2 c = reinterpret_cast<const float2*>(p.A)[i];
      v = (p.a_kind == A_COMPLEX_ABS2) ? c.x * c.x + c.y * c.y : hypotf(c.x, c.y);
    }
    if (p.inverse) {
      if (p.offset) v = __fadd_rn(__fmul_rn(v, sc), off);
      v = contrast_inv(v, p.contrast, p.eps);
    } else {
      v = contrast_fwd(v, p.contrast, p.eps);
      if (p.offset) v = (v - off) / sc;
    }
    p.out[i] = v;
  }
}

template <int KSTEPS, int NL>
static int launch_mel(const MelParams& p0, hipStream_t stream) {
  MelParams p = p0;
  const long long ntiles = (p.rows + ROWS - 1) / ROWS;
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  const int colblocks = (p.N + 127) / 128;
  long long rowblocks = cus / colblocks;
  if (rowblocks < 1) rowblocks = 1;
  if (rowblocks > ntiles) rowblocks = ntiles;
  p.tiles_per_block = (ntiles + rowblocks - 1) / rowblocks;
  rowblocks = (ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
  const size_t lds = sizeof(float) * 2 * ROWS * (size_t)p.rs + 16;
  auto kern = p.dense ? mel_gemm_kernel<KSTEPS, NL, true> : mel_gemm_kernel<KSTEPS, NL, false>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AT_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)rowblocks, (unsigned)colblocks), dim3(THREADS), lds, stream, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int launch_mel_project(const MelParams& p0, hipStream_t s) {
  MelParams p = p0;
  const int K = p.K, N = p.N;
  const long long rows = p.rows;
  // LDS row stride (floats).  hipcc fuses the A-fragment reads of a j-pair into ds_read2_b64, which is
  // serviced in 16-lane groups over 32 banks: consecutive rows must step by 8 B (mod 128 B) => rs = 32q + 2
  p.rs = ((K - 2 + 31) / 32) * 32 + 2;
  p.tiles_per_block = 1;
  if (K > 576 || K < 16) {
    long long total = rows * (long long)N;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mel_gemm_simple_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
  }
  const int nl = (K + 127) / 128;  // 1..5
  // largest K_main = 4*KSTEPS <= K from {512, 256, 128, 64, 32, 16}
#define AT_MEL_CASE(KS)                                        \
  switch (nl) {                                                \
    case 1: return launch_mel<KS, 1>(p, s);                    \
    case 2: return launch_mel<KS, 2>(p, s);                    \
    case 3: return launch_mel<KS, 3>(p, s);                    \
    case 4: return launch_mel<KS, 4>(p, s);                    \
    default: return launch_mel<KS, 5>(p, s);                   \
  }
  if (K >= 512) { AT_MEL_CASE(128) }
  if (K >= 256) { AT_MEL_CASE(64) }
  if (K >= 128) { AT_MEL_CASE(32) }
  if (K >= 64) { AT_MEL_CASE(16) }
  if (K >= 32) { AT_MEL_CASE(8) }
  { AT_MEL_CASE(4) }
#undef AT_MEL_CASE
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_mel_project(const void* A, int a_kind, int64_t rows, int64_t lda, int K, const float* bank, int ldb, int N,
                   int contrast, int inverse, const float* offset, const float* scale, float eps, float* out,
                   int64_t ld_out, int64_t T_transposed, void* stream) {
  if (rows < 0 || K <= 0 || N <= 0) return AT_EINVAL;
  if (rows == 0) return AT_OK;
  if (!A || !bank || !out) return AT_EINVAL;
  if (a_kind < 0 || a_kind > 3 || contrast < 0 || contrast > 3) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (inverse && a_kind != A_REAL) return AT_EINVAL;
  MelParams p;
  p.A = A; p.Bm = bank; p.out = out; p.offset = offset; p.scale = scale;
  p.rows = rows; p.lda = lda; p.ld_out = ld_out; p.T = T_transposed;
  p.K = K; p.N = N; p.ldb = ldb; p.a_kind = a_kind; p.contrast = contrast; p.inverse = inverse; p.eps = eps;
  p.a_block_offset = nullptr;
  p.dense = 0;
  return launch_mel_project(p, (hipStream_t)stream);
}

int at_mag_pointwise(const void* A, int a_kind, int64_t n, int contrast, int inverse, const float* offset,
                     const float* scale, float eps, float* out, void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!A || !out) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  PointParams p = {A, out, offset, scale, n, a_kind, contrast, inverse, eps};
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mag_pointwise_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
