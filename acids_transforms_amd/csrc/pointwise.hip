// pointwise.hip -- small elementwise kernels around the spectral path.
//   angle            x_fft.angle()                 reference stft.py:103, dgt.py:69, dgt.py:336   (K2)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "fastmath.h"
#include <stdint.h>
#include <type_traits>

#include "../../include/acids_hip.h"
#include "variants.h"

namespace at_hip {

__global__ void angle_kernel(const float2* __restrict__ x, long long n, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float2 v = x[i];
    out[i] = fast_atan2f(v.y, v.x);
  }
}

// ---------------------------------------------------------------------------
// global statistics of contrast(|x|) (or of a real tensor): min, max, sum, sum of squares.
//   Normalize.scale_data          reference norm.py:25-38
//   Magnitude.scale_data          reference spectral_repr.py:242-245 (stats of contrast(|x|), no mel)
// Two launches: per-block partials, then one block folds them.  Sums are kept in fp64.
// ---------------------------------------------------------------------------
struct StatsParams {
  const void* A;
  long long n;
  int a_kind;    // 0 complex |.|, 1 complex |.|^2, 2 real, 3 |real|
  int contrast;  // 0 none, 1 log1p, 2 log, 3 log10
  float eps;
  double* partial;  // blocks x 4
  int nblocks;
};

__device__ __forceinline__ float stats_value(const StatsParams& p, long long i) {
  float v;
  if (p.a_kind >= 2) {
    v = reinterpret_cast<const float*>(p.A)[i];
    if (p.a_kind == 3) v = fabsf(v);
  } else {
    float2 c = reinterpret_cast<const float2*>(p.A)[i];
    v = (p.a_kind == 1) ? c.x * c.x + c.y * c.y : hypotf(c.x, c.y);
  }
  switch (p.contrast) {
    case 1: return logf(1.0f + v);
    case 2: return logf(fmaxf(v, p.eps));
    case 3: return log10f(fmaxf(v, p.eps));
    default: return v;
  }
}

__device__ __forceinline__ void block_fold(float mn, float mx, double s, double ss, double* dst) {
  __shared__ float s_mn[4], s_mx[4];
  __shared__ double s_s[4], s_ss[4];
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    s += __shfl_xor(s, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_mn[w] = mn; s_mx[w] = mx; s_s[w] = s; s_ss[w] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      mn = fminf(mn, s_mn[i]); mx = fmaxf(mx, s_mx[i]); s += s_s[i]; ss += s_ss[i];
    }
    dst[0] = mn; dst[1] = mx; dst[2] = s; dst[3] = ss;
  }
}

__global__ __launch_bounds__(256) void stats_partial_kernel(StatsParams p) {
  float mn = INFINITY, mx = -INFINITY;
  double s = 0.0, ss = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x) {
    float v = stats_value(p, i);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    s += (double)v;
    ss += (double)v * (double)v;
  }
  block_fold(mn, mx, s, ss, p.partial + 4 * blockIdx.x);
}

__global__ __launch_bounds__(256) void stats_final_kernel(const double* partial, int nblocks, double* out4) {
  float mn = INFINITY, mx = -INFINITY;
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
    mn = fminf(mn, (float)partial[4 * i]);
    mx = fmaxf(mx, (float)partial[4 * i + 1]);
    s += partial[4 * i + 2];
    ss += partial[4 * i + 3];
  }
  block_fold(mn, mx, s, ss, out4);
}

// (x - offset) / scale   and   x * scale + offset      reference norm.py:40-44
__global__ void affine_kernel(const float* __restrict__ x, long long n, const float* offset, const float* scale,
                              int inverse, float* __restrict__ out) {
  const float off = *offset, sc = *scale;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = inverse ? __fadd_rn(__fmul_rn(x[i], sc), off) : (x[i] - off) / sc;
}

// Cartesian (reference spectral_repr.py:403-428): normalise(x.real) and normalise(x.imag) stacked on dim -2, and back.
// One pass over the spectrum either way: (rows, F) complex64 <-> (rows, 2, F) float32.  A null offset: no Normalize
// on that half.
// (row, bin) of the flat element index are carried along the grid stride (one division per thread, not per element)
template <bool WIDE>
__global__ __launch_bounds__(256) void cartesian_pack_kernel(const float2* __restrict__ x, long long rows, int F,
                                                             const float* re_off, const float* re_sc, const float* im_off,
                                                             const float* im_sc, float* __restrict__ out) {
  const float ro = re_off ? *re_off : 0.f, rs = re_off ? *re_sc : 1.f;
  const float io = im_off ? *im_off : 0.f, is = im_off ? *im_sc : 1.f;
  using Idx = typename std::conditional<WIDE, unsigned long long, unsigned>::type;
  const Idx total = (Idx)rows * (Idx)F;
  const Idx stride = (Idx)gridDim.x * 256;
  const Idx dr = stride / (Idx)F, df = stride - dr * (Idx)F;
  Idx i = (Idx)blockIdx.x * 256 + threadIdx.x;
  Idx r = i / (Idx)F, f = i - r * (Idx)F;
  for (; i < total; i += stride) {
    const float2 v = x[i];
    float* dst = out + (2 * (unsigned long long)r) * F + f;
    dst[0] = re_off ? (v.x - ro) / rs : v.x;
    dst[F] = im_off ? (v.y - io) / is : v.y;
    r += dr;
    f += df;
    if (f >= (Idx)F) {
      f -= (Idx)F;
      ++r;
    }
  }
}

// The same with ROWS-PER-BLOCK chunks and the block's wavefronts in lockstep, for rows that are not whole 64-byte
// segments (F = 513): in the grid-stride form above a wavefront's 256 bytes of a real (or imaginary) row begin and end
// inside a segment, and the segment's other half comes from the neighbouring block -- on another XCD, behind another
// L2 -- so the memory side sees two partial writes per block edge (phase_repr.hip has the counters of the same effect in
// the scans).  Here a block owns kPackRows consecutive frames (2 kPackRows output rows, contiguous), its wavefronts walk
// them together (a barrier per 256 elements), and only the chunk's two ends are shared with other blocks.
constexpr int kPackRows = 8;
__global__ __launch_bounds__(256) void cartesian_pack_rows_kernel(const float2* __restrict__ x, long long rows, int F,
                                                                  const float* re_off, const float* re_sc, const float* im_off,
                                                                  const float* im_sc, float* __restrict__ out) {
  const float ro = re_off ? *re_off : 0.f, rs = re_off ? *re_sc : 1.f;
  const float io = im_off ? *im_off : 0.f, is = im_off ? *im_sc : 1.f;
  const long long r0 = (long long)blockIdx.x * kPackRows;
  const int nr = (int)(rows - r0 < kPackRows ? rows - r0 : kPackRows);
  const int n = nr * F;
  const float2* src = x + r0 * F;
  float* dst0 = out + 2 * r0 * F;
  const int dr = 256 / F, df = 256 - dr * F;
  int r = (int)threadIdx.x / F, f = (int)threadIdx.x - r * F;
  for (int i = threadIdx.x; i < ((n + 255) & ~255); i += 256) {
    if (i < n) {
      const float2 v = src[i];
      float* dst = dst0 + (long long)(2 * r) * F + f;
      dst[0] = re_off ? (v.x - ro) / rs : v.x;
      dst[F] = im_off ? (v.y - io) / is : v.y;
    }
    r += dr;
    f += df;
    if (f >= F) {
      f -= F;
      ++r;
    }
    __syncthreads();
  }
}

template <bool WIDE>
__global__ __launch_bounds__(256) void cartesian_unpack_kernel(const float* __restrict__ y, long long rows, int F,
                                                               const float* re_off, const float* re_sc,
                                                               const float* im_off, const float* im_sc,
                                                               float2* __restrict__ out) {
  const float ro = re_off ? *re_off : 0.f, rs = re_off ? *re_sc : 1.f;
  const float io = im_off ? *im_off : 0.f, is = im_off ? *im_sc : 1.f;
  using Idx = typename std::conditional<WIDE, unsigned long long, unsigned>::type;
  const Idx total = (Idx)rows * (Idx)F;
  const Idx stride = (Idx)gridDim.x * 256;
  const Idx dr = stride / (Idx)F, df = stride - dr * (Idx)F;
  Idx i = (Idx)blockIdx.x * 256 + threadIdx.x;
  Idx r = i / (Idx)F, f = i - r * (Idx)F;
  for (; i < total; i += stride) {
    const float* src = y + (2 * (unsigned long long)r) * F + f;
    const float re = src[0], im = src[F];
    out[i] = make_float2(re_off ? __fadd_rn(__fmul_rn(re, rs), ro) : re, im_off ? __fadd_rn(__fmul_rn(im, is), io) : im);
    r += dr;
    f += df;
    if (f >= (Idx)F) {
      f -= (Idx)F;
      ++r;
    }
  }
}

// ---------------------------------------------------------------------------
// OverlapAdd streaming state (reference transforms/oadd.py)
//   forward :69-74, 33-42  buf = [history | chunk | zero pad]; history <- chunk[-keep:]      (K6)
//   invert  :90-104        rec = [tail | 0] + sum_i frames[i] shifted by i*hop;
//                          out = rec[:-keep] / gain; tail <- rec[-keep:]                      (K7)
// ---------------------------------------------------------------------------
__global__ void oadd_forward_kernel(const float* __restrict__ x, const float* __restrict__ hist_in, int S, long long C,
                                    int keep, long long buf_len, float* __restrict__ buf, float* __restrict__ hist_out) {
  const long long total = (long long)S * buf_len;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i / buf_len, p = i - s * buf_len;
    float v = 0.f;
    if (p < keep) v = hist_in ? hist_in[s * keep + p] : 0.f;
    else if (p < keep + C) v = x[s * C + (p - keep)];
    buf[i] = v;
    if (p >= C && p < C + keep) hist_out[s * keep + (p - C)] = v;  // last `keep` samples of [hist | chunk]
  }
}

// One workgroup per stream.  The carried tail is staged in LDS before anything is written, so tail_out may be the
// very buffer tail_in points to (the streaming session updates its state in place, no copy afterwards).
__global__ __launch_bounds__(256) void oadd_invert_kernel(const float* frames, const float* tail_in, int S, int n,
                                                          int n_fft, int hop, int keep, const float* gain, float* out,
                                                          float* tail_out) {
  extern __shared__ float tail_lds[];
  const long long rec_len = (long long)(n - 1) * hop + n_fft;
  const long long out_len = rec_len - keep;
  const float g = *gain;
  for (long long s = blockIdx.x; s < S; s += gridDim.x) {
    for (int j = threadIdx.x; j < keep; j += blockDim.x) tail_lds[j] = tail_in ? tail_in[s * keep + j] : 0.f;
    __syncthreads();
    for (long long p = threadIdx.x; p < rec_len; p += blockDim.x) {
      float acc = p < keep ? tail_lds[p] : 0.f;
      long long t_hi = p / hop;
      if (t_hi > n - 1) t_hi = n - 1;
      long long t_lo = (p >= n_fft) ? (p - n_fft + hop) / hop : 0;
      for (long long t = t_lo; t <= t_hi; ++t) {  // ascending frame order, like the reference's += loop
        const long long o = p - t * hop;
        if (o >= 0 && o < n_fft) acc += frames[(s * n + t) * n_fft + o];
      }
      if (p < out_len) out[s * out_len + p] = acc / g;
      else tail_out[s * keep + (p - out_len)] = acc;
    }
    __syncthreads();
  }
}

// Streaming input state kept IN PLACE: buf (S, buf_len) holds [history | previous chunk | pad].  One step moves the last
// `keep` samples of [history | previous chunk] to the front and writes the new chunk behind them -- what
// OverlapAdd.forward does with its input_buffer (oadd.py:33-42, 69-74), for any chunk length C >= 1 (a hop-sized chunk
// included: the regions overlap then, hence the LDS staging).  One workgroup per stream.
__global__ __launch_bounds__(256) void oadd_push_kernel(const float* __restrict__ x, int S, long long C, int keep,
                                                        long long buf_len, float* buf) {
  extern __shared__ float win_lds[];
  for (long long s = blockIdx.x; s < S; s += gridDim.x) {
    float* b = buf + s * buf_len;
    for (int j = threadIdx.x; j < keep; j += blockDim.x) win_lds[j] = b[C + j];
    __syncthreads();
    for (int j = threadIdx.x; j < keep; j += blockDim.x) b[j] = win_lds[j];
    for (long long j = threadIdx.x; j < C; j += blockDim.x) b[keep + j] = x[s * C + j];
    __syncthreads();
  }
}

// Griffin-Lim phase update (torchaudio.functional.griffinlim as called at reference stft.py:174-178):
//   angles = rebuilt - momentum' * tprev;  angles /= (|angles| + 1e-16);  X = mag * angles
// tprev == nullptr on the first iteration (the reference starts from tprev = 0).
__global__ void griffinlim_update_kernel(const float* __restrict__ mag, const float2* __restrict__ rebuilt,
                                         const float2* __restrict__ tprev, float mom, long long n,
                                         float2* __restrict__ X) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float2 a = rebuilt[i];
    if (tprev) {
      const float2 t = tprev[i];
      a.x -= mom * t.x;
      a.y -= mom * t.y;
    }
    const float d = hypotf(a.x, a.y) + 1e-16f;
    const float m = mag[i];
    X[i] = make_float2(m * (a.x / d), m * (a.y / d));
  }
}

// X = mag * z for a complex z (initial random "angles")
__global__ void scale_complex_kernel(const float* __restrict__ mag, const float2* __restrict__ z, long long n,
                                     float2* __restrict__ X) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float m = mag[i];
    X[i] = make_float2(m * z[i].x, m * z[i].y);
  }
}

static inline unsigned grid_for(long long n, int block) {
  long long b = (n + block - 1) / block;
  if (b > 256 * 8) b = 256 * 8;  // grid-stride above 8 blocks per CU
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_angle(const float* x_complex, int64_t n, float* out, void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!x_complex || !out) return AT_EINVAL;
  hipLaunchKernelGGL(angle_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)x_complex,
                     (long long)n, out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}


#define AT_STATS_MAX_BLOCKS 1024

size_t at_stats_workspace_bytes(void) { return sizeof(double) * 4 * AT_STATS_MAX_BLOCKS; }

int at_stats(const void* A, int a_kind, int64_t n, int contrast, float eps, double* out4, void* workspace,
             size_t workspace_bytes, void* stream) {
  if (n <= 0 || !A || !out4) return AT_EINVAL;
  if (a_kind < 0 || a_kind > 3 || contrast < 0 || contrast > 3) return AT_EINVAL;
  if (!workspace || workspace_bytes < at_stats_workspace_bytes()) return AT_EWORKSPACE;
  long long blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > AT_STATS_MAX_BLOCKS) blocks = AT_STATS_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  StatsParams p = {A, (long long)n, a_kind, contrast, eps, (double*)workspace, (int)blocks};
  hipLaunchKernelGGL(stats_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace,
                     (int)blocks, out4);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_griffinlim_update(const float* mag, const float* rebuilt_complex, const float* tprev_complex_or_null,
                         float momentum_over_1p, int64_t n, float* X_complex, void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!mag || !rebuilt_complex || !X_complex) return AT_EINVAL;
  hipLaunchKernelGGL(griffinlim_update_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, mag,
                     (const float2*)rebuilt_complex, (const float2*)tprev_complex_or_null, momentum_over_1p, (long long)n,
                     (float2*)X_complex);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_scale_complex(const float* mag, const float* z_complex, int64_t n, float* X_complex, void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!mag || !z_complex || !X_complex) return AT_EINVAL;
  hipLaunchKernelGGL(scale_complex_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, mag,
                     (const float2*)z_complex, (long long)n, (float2*)X_complex);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_affine(const float* x, int64_t n, const float* offset, const float* scale, int inverse, float* out,
              void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!x || !offset || !scale || !out) return AT_EINVAL;
  hipLaunchKernelGGL(affine_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, offset,
                     scale, inverse, out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}


int at_cartesian_pack(const float* x_complex, int64_t rows, int F, const float* re_offset, const float* re_scale,
                      const float* im_offset, const float* im_scale, float* stacked, void* stream) {
  if (rows < 0 || F <= 0) return AT_EINVAL;
  if (rows == 0) return AT_OK;
  if (!x_complex || !stacked) return AT_EINVAL;
  if ((re_offset == nullptr) != (re_scale == nullptr) || (im_offset == nullptr) != (im_scale == nullptr)) return AT_EINVAL;
  const long long n = (long long)rows * F;
  const unsigned grid = grid_for(n, 256) * 4;      // grid_for caps at 8 blocks per CU: one element per thread and trip here
  if ((F * 4) % 64 != 0 && F >= 64 && rows >= 64 && (rows + kPackRows - 1) / kPackRows < (1LL << 31) &&
      !dev_env("ACIDS_CARTESIAN_FLAT")) {
    hipLaunchKernelGGL(cartesian_pack_rows_kernel, dim3((unsigned)((rows + kPackRows - 1) / kPackRows)), dim3(256), 0,
                       (hipStream_t)stream, (const float2*)x_complex, (long long)rows, F, re_offset, re_scale, im_offset,
                       im_scale, stacked);
    return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
  }
  if (n < (1LL << 32))
    hipLaunchKernelGGL(cartesian_pack_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float2*)x_complex,
                       (long long)rows, F, re_offset, re_scale, im_offset, im_scale, stacked);
  else
    hipLaunchKernelGGL(cartesian_pack_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float2*)x_complex,
                       (long long)rows, F, re_offset, re_scale, im_offset, im_scale, stacked);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_cartesian_unpack(const float* stacked, int64_t rows, int F, const float* re_offset, const float* re_scale,
                        const float* im_offset, const float* im_scale, float* out_complex, void* stream) {
  if (rows < 0 || F <= 0) return AT_EINVAL;
  if (rows == 0) return AT_OK;
  if (!stacked || !out_complex) return AT_EINVAL;
  if ((re_offset == nullptr) != (re_scale == nullptr) || (im_offset == nullptr) != (im_scale == nullptr)) return AT_EINVAL;
  const long long n = (long long)rows * F;
  const unsigned grid = grid_for(n, 256) * 4;
  if (n < (1LL << 32))
    hipLaunchKernelGGL(cartesian_unpack_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, stacked, (long long)rows,
                       F, re_offset, re_scale, im_offset, im_scale, (float2*)out_complex);
  else
    hipLaunchKernelGGL(cartesian_unpack_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, stacked, (long long)rows,
                       F, re_offset, re_scale, im_offset, im_scale, (float2*)out_complex);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_oadd_forward(const float* x, const float* hist_in_or_null, int S, int64_t C, int keep, int64_t buf_len,
                    float* buf, float* hist_out, void* stream) {
  if (S < 0 || C <= 0 || keep < 0 || buf_len < keep + C) return AT_EINVAL;
  // C < keep (e.g. one hop per step) is an extension: the reference's history slice silently shortens there and its
  // next call fails (oadd.py:41); here hist_out is always the last `keep` samples of [history | chunk]
  if (S == 0) return AT_OK;
  if (!x || !buf || !hist_out) return AT_EINVAL;
  hipLaunchKernelGGL(oadd_forward_kernel, dim3(grid_for((long long)S * buf_len, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     hist_in_or_null, S, (long long)C, keep, (long long)buf_len, buf, hist_out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_oadd_invert(const float* frames, const float* tail_in_or_null, int S, int n, int n_fft, int hop, int keep,
                   const float* gain, float* out, float* tail_out, void* stream) {
  if (S < 0 || n <= 0 || n_fft <= 0 || hop <= 0 || keep < 0) return AT_EINVAL;
  if ((long long)(n - 1) * hop + n_fft < keep) return AT_EINVAL;
  if (S == 0) return AT_OK;
  if (!frames || !gain || !out || !tail_out) return AT_EINVAL;
  if ((size_t)keep * sizeof(float) > 64 * 1024) return AT_EUNSUPPORTED;
  hipLaunchKernelGGL(oadd_invert_kernel, dim3(S < 65535 ? S : 65535), dim3(256), (size_t)keep * sizeof(float),
                     (hipStream_t)stream, frames, tail_in_or_null, S, n, n_fft, hop, keep, gain, out, tail_out);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_oadd_push(const float* x, int S, int64_t C, int keep, int64_t buf_len, float* buf, void* stream) {
  if (S < 0 || C <= 0 || keep < 0 || buf_len < keep + C) return AT_EINVAL;
  if (S == 0) return AT_OK;
  if (!x || !buf) return AT_EINVAL;
  if ((size_t)keep * sizeof(float) > 64 * 1024) return AT_EUNSUPPORTED;
  hipLaunchKernelGGL(oadd_push_kernel, dim3(S < 65535 ? S : 65535), dim3(256), (size_t)keep * sizeof(float),
                     (hipStream_t)stream, x, S, (long long)C, keep, (long long)buf_len, buf);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
