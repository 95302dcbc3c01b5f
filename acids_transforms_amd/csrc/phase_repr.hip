// phase_repr.hip -- the phase side of the spectral representations: unwrap, instantaneous frequency
// (finite differences of the unwrapped phase) and its inverse (integration along time).
//
// Replaces utils/misc.py:12-26 (unwrap), :65-81 (fdiff_*), :83-104 (fint_*), and the per-row scalings
// of IF.get_if / IF.invert (transforms/spectral_repr.py:318-335, 360-373), fused with the angle of the
// complex spectrum in front and the Normalize affine behind so that the spectrum is read once.
//
// All of these are scans along the frame axis of a (B, T, F) tensor.  One thread owns one (clip, bin) column
// and walks it in frame order -- the reference's arithmetic is sequential in t (torch.cumsum on CPU
// accumulates in double and rounds every prefix to float; fint_central is a Python loop), so the walk
// reproduces it operation by operation (built with -ffp-contract=off).  Adjacent threads own adjacent bins:
// every step of a wavefront reads/writes 64 consecutive elements of one frame row.  Loads do not depend on
// the recurrence and are unrolled ahead of it.  HBM-bound: 8 (complex) or 4 (phase) bytes in, 4 out per bin.
#include <hip/hip_runtime.h>
#include "fastmath.h"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/acids_hip.h"
#include "mel_gemm.h"   // C_* codes, banded_contrast_fwd
#include "variants.h"

namespace at_hip {

constexpr float kPi = 3.14159265358979323846f;        // float(torch.pi)
constexpr float kTwoPi = 6.28318530717958647692f;     // float(2 * torch.pi)

enum { SCAN_UNWRAP = 0, SCAN_IF_FORWARD = 1, SCAN_IF_BACKWARD = 2, SCAN_IF_CENTRAL = 3, SCAN_ANGLE = 4 };

struct ScanParams {
  const float2* X;       // complex spectrum, or
  const float* phase;    // wrapped phase (exactly one of the two)
  float* out;
  long long B, T, F;
  const float* window;   // optional per-frame weight (IF weighted), T floats
  const float* offset;   // optional Normalize affine (device scalars)
  const float* scale;
  int bare;              // 1: plain fdiff_* of a real signal (no unwrap, no per-row division): utils/misc.py:65-81
  long long ld_out;      // floats between consecutive frames of `out` (F, or 2 F inside a stacked (.., T, 2, F) tensor)
  // MAG (PolarIF.forward in one pass, clip-per-block layout only): the banded magnitude projection of the same rows,
  // normalise(contrast(|X| @ bank)), rows of ld_out floats at mag_out.  The bank by filter: first bin, bins, offset of its
  // weights in band_w (the dense bank's values from its first to its last non-zero row, ascending).
  float* mag_out;
  const int* band_start;
  const int* band_len;
  const int* band_off;
  const float* band_w;
  int n_w;               // floats in band_w
  int row_floats;        // LDS row: F rounded up to a multiple of 4, + 4
  int contrast;
  float eps;
  const float* mag_offset;
  const float* mag_scale;
};

// the correction torch's unwrap adds for one frame-to-frame jump (utils/misc.py:19-24)
__device__ __forceinline__ float unwrap_correction(float jump) {
  // torch.remainder(x, 2 pi) = fmod, then + 2 pi when the result is negative.  Angles differ by less than 2 pi,
  // so x = jump + pi lies in (-pi, 3 pi): there fmod is x itself or x - 2 pi, and that subtraction is exact
  // (Sterbenz: 2 pi <= x <= 4 pi).  Anything else (arbitrary real input) takes the library fmod.
  // As selects rather than a chain of branches: the same three cases, and the library call only outside
  // (-2 pi, 4 pi) -- never for angles.  (Timing-neutral: the scans wait on their column walk, not on instructions.)
  const float x = jump + kPi;
  float r = (x >= kTwoPi) ? x - kTwoPi : x;
  if (!(x > -kTwoPi && x < 2.0f * kTwoPi)) r = fmodf(x, kTwoPi);
  r = (r < 0.0f) ? r + kTwoPi : r;
  float folded = r - kPi;
  if (folded == -kPi && jump > 0.0f) folded = kPi;
  const float corr = folded - jump;
  return (fabsf(jump) < kPi) ? 0.0f : corr;
}

constexpr int kRowsAhead = 8;   // rows requested before the recurrence consumes them (per thread: 64 B in flight)

// Input kind, weighting and normalisation are template flags: with them as run-time tests the frame loop is
// full of branches and every load is followed by vmcnt(0) -- one row in flight per thread, 3.1 TB/s for the
// plain angle against 4.9 TB/s for the same bytes read elementwise.
//
// NC = 1: flattened (clip, bin) columns, 256 per block (any shape).  NC = 2 / 4: ONE BLOCK PER CLIP, thread j walks
// the NC columns j + k H, H = ceil(F / NC), and the block's wavefronts advance in lockstep (a barrier per batch of rows).
// Rows of F = 513 floats start 4 bytes further into their 64-byte segment than the row before, so the run of columns
// a wavefront stores begins and ends inside a segment on almost every row, and whoever owns the neighbouring columns
// completes that segment at some other time: the memory side sees partial writes (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B =
// 3.6 M per launch against 22.7 M whole ones) and the walk runs at 3.4 TB/s where the same walk over 512-float rows runs
// at 5.1 TB/s (tools/scan_align_probe.py; misaligned LOADS cost nothing).  What did not help: staging the block's rows
// through LDS to re-cut the stores on line boundaries (2.1 M partial writes left: the block edges); one block per clip
// without the barrier (3.1 M: the wavefronts drift).  With the whole clip in one block there are no block edges inside a
// row, the end of row t meets the start of row t + 1 in the same block, and the barrier keeps both halves of every shared
// segment within L2's reach.  Two columns per thread because 513 columns are 8.02 wavefronts: 257 threads = 5 wavefronts
// per clip, 1024 clips resident at once.
template <int MODE, bool CPLX, bool WIN, bool NORM, int NC, bool MAG = false>
__global__ __launch_bounds__(NC > 1 ? 1024 : 256) void phase_scan_kernel(ScanParams p) {
  static_assert(!MAG || (NC > 1 && CPLX && MODE != SCAN_ANGLE && MODE != SCAN_UNWRAP), "MAG: the IF scans of a complex spectrum, one block per clip");
  const long long T = p.T, F = p.F;
  long long b, fk[NC];
  bool on[NC];
  if (NC == 1) {
    const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= p.B * F) return;
    b = col / F;
    fk[0] = col - b * F;
    on[0] = true;
  } else {
    b = blockIdx.x;
    const long long H = (F + NC - 1) / NC;
    // the block is H rounded up to whole wavefronts: the surplus threads stay alive (every thread of the block must
    // reach the barriers of the row loop), walk column 0 and store nothing
    const bool live = (long long)threadIdx.x < H;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const long long f = threadIdx.x + k * H;
      on[k] = live && f < F;
      fk[k] = on[k] ? f : (live ? (long long)threadIdx.x : 0);     // a column past the end walks a valid column again and stores nothing
    }
  }
  float off = 0.f, sc = 1.f;
  if (NORM) {
    off = *p.offset;
    sc = *p.scale;
  }
  const bool div = !p.bare;
  float* out_row0 = p.out + b * T * p.ld_out;
  const long long ldo = p.ld_out;
  auto emit = [&](int k, long long t, float v) {
    if (WIN) v = p.window[t] * v;
    if (NORM) v = (v - off) / sc;
    if (NC == 1 || on[k]) out_row0[t * ldo + fk[k]] = v;
  };
  using In = typename std::conditional<CPLX, float2, float>::type;
  const In* src = (CPLX ? reinterpret_cast<const In*>(p.X) : reinterpret_cast<const In*>(p.phase)) + b * T * F;
  auto to_phase = [](In v) -> float {
    if constexpr (CPLX) return fast_atan2f(v.y, v.x);
    else return v;
  };
  double acc[NC];                   // torch.cumsum's accumulator on CPU
  float raw_prev[NC];
  float u_prev[NC];                 // unwrapped phase of frame t-1
  float u_prev2[NC];                // ... of frame t-2
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    acc[k] = 0.0;
    raw_prev[k] = u_prev[k] = u_prev2[k] = 0.f;
  }
  auto step = [&](int k, long long t, In v) {
    const float raw = to_phase(v);
    if (MODE == SCAN_ANGLE) {
      emit(k, t, raw);
      return;
    }
    if (t == 0) {
      if (MODE == SCAN_UNWRAP || MODE == SCAN_IF_CENTRAL) emit(k, 0, raw);
      if (MODE == SCAN_IF_FORWARD) emit(k, 0, (div && T > 1) ? raw / kPi : raw);   // rows [0, T-2] are divided by pi
      raw_prev[k] = u_prev[k] = raw;
      return;
    }
    if (div) acc[k] += (double)unwrap_correction(raw - raw_prev[k]);
    const float u = raw + (float)acc[k];
    if (MODE == SCAN_UNWRAP) {
      emit(k, t, u);
    } else if (MODE == SCAN_IF_FORWARD) {
      const float d = (u - u_prev[k]) / 2.0f;
      emit(k, t, (div && t < T - 1) ? d / kPi : d);
    } else if (MODE == SCAN_IF_BACKWARD) {
      const float d = (u_prev[k] - u) / 2.0f;         // row t-1; rows >= 1 are divided by -pi
      emit(k, t - 1, (div && t - 1 >= 1) ? d / (-kPi) : d);
    } else if (MODE == SCAN_IF_CENTRAL) {
      if (t >= 2) {                                   // interior rows
        const float d = (u - u_prev2[k]) / 4.0f;
        emit(k, t - 1, div ? d / kTwoPi : d);
      }
    }
    raw_prev[k] = raw;
    u_prev2[k] = u_prev[k];
    u_prev[k] = u;
  };
  if constexpr (MAG) {
    // PolarIF.forward: the rows this block has in registers for the scan also go, as |X|, through LDS into the banded
    // projection -- thread j sums the bands of its NC filters over the batch's rows, in the order the banded walk of
    // mel_banded.hip adds them (ascending bins, fmaf; the walk's zero-weight padding adds nothing), so the magnitude half
    // equals at_mel_project_banded's bit for bit and the spectrum is read once instead of twice.
    extern __shared__ float scan_lds[];
    float* wl = scan_lds;                                   // the bank's weights
    float* rows = scan_lds + ((p.n_w + 3) & ~3);           // kRowsAhead values of |X| per bin
    for (int i = threadIdx.x; i < p.n_w; i += blockDim.x) wl[i] = p.band_w[i];
    int fs[NC], fl[NC], fo[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      fs[k] = on[k] ? p.band_start[fk[k]] : 0;
      fl[k] = on[k] ? p.band_len[fk[k]] : 0;
      fo[k] = on[k] ? p.band_off[fk[k]] : 0;
    }
    float moff = 0.f, minv = 1.f;
    if (p.mag_offset) {
      moff = *p.mag_offset;
      minv = 1.0f / *p.mag_scale;
    }
    float* mag_row0 = p.mag_out + b * T * p.ld_out;
    // Where the 1.85 ms per 1024 x 690 x 513 go (builds with a part removed): the scan's own arithmetic (atan2, unwrap, the
    // double accumulator, two IEEE divisions per bin) 0.8, the walk 0.35, contrast + normalise 0.2, the magnitude stores 0.1.
    // (Requesting the next batch before working on this one costs 32 registers: 121, a fourth block no longer fits a CU
    // and the 1024 clips take two rounds -- 1.85 -> 2.50 ms.)
    for (long long t = 0; t < T; t += kRowsAhead) {
      const int nr = (T - t < kRowsAhead) ? (int)(T - t) : kRowsAhead;
      In v[NC][kRowsAhead];
#pragma unroll
      for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int r = 0; r < kRowsAhead; ++r) v[k][r] = src[(r < nr ? t + r : T - 1) * F + fk[k]];
      // bin-major in LDS: the batch's eight |X| of a bin are 32 contiguous bytes (two 16-byte writes per column, two
      // 16-byte reads per weight)
      static_assert(kRowsAhead == 8, "two float4 per bin");
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (on[k]) {
          float m[kRowsAhead];
#pragma unroll
          for (int r = 0; r < kRowsAhead; ++r) m[r] = __builtin_amdgcn_sqrtf(fmaf(v[k][r].x, v[k][r].x, v[k][r].y * v[k][r].y));
          float4* dst = reinterpret_cast<float4*>(rows) + 2 * fk[k];
          dst[0] = make_float4(m[0], m[1], m[2], m[3]);
          dst[1] = make_float4(m[4], m[5], m[6], m[7]);
        }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < kRowsAhead; ++r)
#pragma unroll
        for (int k = 0; k < NC; ++k)
          if (r < nr) step(k, t + r, v[k][r]);
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        float acc[kRowsAhead];
#pragma unroll
        for (int r = 0; r < kRowsAhead; ++r) acc[r] = 0.f;
        const float4* a = reinterpret_cast<const float4*>(rows) + 2 * fs[k];
        const float* w = wl + fo[k];
        auto mac = [&](float wj, const float4& lo, const float4& hi) {
          acc[0] = fmaf(lo.x, wj, acc[0]);
          acc[1] = fmaf(lo.y, wj, acc[1]);
          acc[2] = fmaf(lo.z, wj, acc[2]);
          acc[3] = fmaf(lo.w, wj, acc[3]);
          acc[4] = fmaf(hi.x, wj, acc[4]);
          acc[5] = fmaf(hi.y, wj, acc[5]);
          acc[6] = fmaf(hi.z, wj, acc[6]);
          acc[7] = fmaf(hi.w, wj, acc[7]);
        };
        int j = 0;
        for (; j + 2 <= fl[k]; j += 2) {       // two weights a trip: their reads are in flight together
          const float w0 = w[j], w1 = w[j + 1];
          const float4 l0 = a[2 * j], h0 = a[2 * j + 1], l1 = a[2 * j + 2], h1 = a[2 * j + 3];
          mac(w0, l0, h0);
          mac(w1, l1, h1);
        }
        if (j < fl[k]) mac(w[j], a[2 * j], a[2 * j + 1]);
        if (on[k]) {
#pragma unroll
          for (int r = 0; r < kRowsAhead; ++r)
            if (r < nr) {
              float m = banded_contrast_fwd(acc[r], p.contrast, p.eps);
              if (p.mag_offset) m = (m - moff) * minv;
              mag_row0[(t + r) * ldo + fk[k]] = m;
            }
        }
      }
      __syncthreads();       // lockstep (below), and the rows are free again
    }
  } else {
  long long t = 0;
  for (; t + kRowsAhead <= T; t += kRowsAhead) {
    In v[NC][kRowsAhead];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
      for (int r = 0; r < kRowsAhead; ++r) v[k][r] = src[(t + r) * F + fk[k]];
#pragma unroll
    for (int r = 0; r < kRowsAhead; ++r)
#pragma unroll
      for (int k = 0; k < NC; ++k) step(k, t + r, v[k][r]);
    // Lockstep.  The wavefronts of a block write neighbouring pieces of the same rows, and the 64-byte segment two of
    // them share is written whole only if both halves reach L2 close together: left to themselves the wavefronts drift
    // apart and the memory side sees two partial writes (3.1 M per launch).  One barrier per batch of rows:
    // 25 k partial writes, no line fetched twice (TCC_EA0_RDREQ 14.0 M -> 11.35 M = the input once), IF.invert 0.84 -> 0.60 ms.
    if (NC > 1) __syncthreads();
  }
  for (; t < T; ++t)
#pragma unroll
    for (int k = 0; k < NC; ++k) step(k, t, src[t * F + fk[k]]);
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    if (MODE == SCAN_IF_BACKWARD) emit(k, T - 1, (div && T > 1) ? u_prev[k] / (-kPi) : u_prev[k]);   // last row = the phase itself
    if (MODE == SCAN_IF_CENTRAL && T > 1) emit(k, T - 1, u_prev[k]);
  }
}

// Phase(unwrap=False): no recurrence along time, so no column walk -- a flat grid-stride pass (the same bytes
// move at 4.7-4.9 TB/s this way against 3.6 TB/s for the walk).
template <bool CPLX>
__global__ __launch_bounds__(256) void phase_angle_kernel(ScanParams p) {
  const long long total = p.B * p.T * p.F;
  float off = 0.f, sc = 1.f;
  const bool norm = p.offset != nullptr;
  if (norm) {
    off = *p.offset;
    sc = *p.scale;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float v;
    if (CPLX) {
      const float2 z = p.X[i];
      v = fast_atan2f(z.y, z.x);
    } else {
      v = p.phase[i];
    }
    if (p.window) v = p.window[(i / p.F) % p.T] * v;
    if (norm) v = (v - off) / sc;
    if (p.ld_out == p.F) {
      p.out[i] = v;
    } else {
      const long long row = i / p.F;
      p.out[row * p.ld_out + (i - row * p.F)] = v;
    }
  }
}

// ---- integration (IF.invert) -------------------------------------------------------------------------
struct IntParams {
  const float* y;        // (B, T, F) instantaneous frequency (normalised when offset/scale are given)
  float* out;            // (B, T, F) phase -- or complex64 mag * exp(i phase) when `mag` is given
  long long B, T, F;
  const float* offset;
  const float* scale;
  long long ld_y;        // floats between consecutive frames of y (F, or 2 F inside a stacked tensor)
  const float* mag;      // optional (B, T, F) contiguous magnitudes
};

// the integrated phase of one bin goes out as it is, or as mag * exp(i phase) (PolarIF.invert in one pass).  The
// magnitude is an argument: the callers request it with the rows of a batch -- loaded here, one per element, it queued
// behind the previous element's store (loads and stores share vmcnt and return in order).
template <bool POLAR>
__device__ __forceinline__ float mag_at(const IntParams& p, long long idx) {
  if constexpr (POLAR) return p.mag[idx];
  return 0.f;
}
template <bool POLAR>
__device__ __forceinline__ void put_phase(const IntParams& p, long long idx, float ph, float m) {
  if constexpr (POLAR) {
    float sn, cs;
    fast_sincosf(ph, sn, cs);
    reinterpret_cast<float2*>(p.out)[idx] = make_float2(m * cs, m * sn);
  } else {
    p.out[idx] = ph;
  }
}

// NC: columns per thread, as in phase_scan_kernel (NC > 1: one block per clip; forward / backward only)
template <int METHOD, bool NORM, bool POLAR, int NC>   // SCAN_IF_*; `rescale` = 0 gives the bare fint_* of utils/misc.py
__global__ __launch_bounds__(NC > 1 ? 1024 : 256) void phase_integrate_kernel(IntParams p, int rescale) {
  const long long T = p.T, F = p.F;
  long long b, fk[NC];
  bool on[NC];
  if (NC == 1) {
    const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= p.B * F) return;
    b = col / F;
    fk[0] = col - b * F;
    on[0] = true;
  } else {
    b = blockIdx.x;
    const long long H = (F + NC - 1) / NC;
    const bool live = (long long)threadIdx.x < H;       // surplus threads stay for the barriers, see phase_scan_kernel
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const long long f = threadIdx.x + k * H;
      on[k] = live && f < F;
      fk[k] = on[k] ? f : (live ? (long long)threadIdx.x : 0);
    }
  }
  const long long f = fk[0];
  const long long base = b * T * F + f;
  float off = 0.f, sc = 1.f;
  if (NORM) {
    off = *p.offset;
    sc = *p.scale;
  }
  const float* src = p.y + b * T * p.ld_y + f;
  const long long ldy = p.ld_y;
  // de-normalised, re-scaled value of input row t (spectral_repr.py:362-370)
  auto prep = [&](long long t, float v) {
    if (NORM) v = v * sc + off;
    if (rescale) {
      if (METHOD == SCAN_IF_FORWARD && t < T - 1) v = v * kPi;
      if (METHOD == SCAN_IF_BACKWARD && t >= 1) v = v * (-kPi);
      if (METHOD == SCAN_IF_CENTRAL && t >= 1 && t < T - 1) v = v * kTwoPi;
    }
    return v;
  };
  if (METHOD == SCAN_IF_FORWARD || METHOD == SCAN_IF_BACKWARD) {
    // forward: rows >= 1 doubled, running sum from row 0; backward: the mirror image from row T-1
    constexpr bool FWD = (METHOD == SCAN_IF_FORWARD);
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    const float* src_row0 = p.y + b * T * ldy;
    const long long out_row0 = b * T * F;
    auto step = [&](int k, long long s, float v, float m) {   // s = position in scan order, row = FWD ? s : T-1-s
      const long long row = FWD ? s : T - 1 - s;
      v = prep(row, v);
      if (s >= 1) v = v * 2.0f;
      acc[k] += (double)v;
      if (NC == 1 || on[k]) put_phase<POLAR>(p, out_row0 + row * F + fk[k], (float)acc[k], m);
    };
    long long s = 0;
    for (; s + kRowsAhead <= T; s += kRowsAhead) {
      float v[NC][kRowsAhead], m[NC][kRowsAhead];
#pragma unroll
      for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int r = 0; r < kRowsAhead; ++r) {
          const long long row = FWD ? s + r : T - 1 - s - r;
          v[k][r] = src_row0[row * ldy + fk[k]];
          m[k][r] = mag_at<POLAR>(p, out_row0 + row * F + fk[k]);
        }
#pragma unroll
      for (int r = 0; r < kRowsAhead; ++r)
#pragma unroll
        for (int k = 0; k < NC; ++k) step(k, s + r, v[k][r], m[k][r]);
      if (NC > 1) __syncthreads();     // lockstep: see phase_scan_kernel
    }
    for (; s < T; ++s)
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const long long row = FWD ? s : T - 1 - s;
        step(k, s, src_row0[row * ldy + fk[k]], mag_at<POLAR>(p, out_row0 + row * F + fk[k]));
      }
  } else if constexpr (NC > 1) {
    // fint_central (utils/misc.py:96-104) in the clip-per-block layout (round 5): the same statements as the one-column
    // form below, NC columns per thread, the block's wavefronts in lockstep so that whole rows leave together (what
    // made the forward / backward scans 1.5x faster at F = 513, see phase_scan_kernel).
    const float* src_row0 = p.y + b * T * ldy;
    const long long out_row0 = b * T * F;
    auto zk = [&](int k, long long t) { return prep(t, src_row0[t * ldy + fk[k]]); };
    auto putk = [&](int k, long long row, float ph) {
      if (on[k]) p.out[out_row0 + row * F + fk[k]] = ph;
    };
    constexpr int RA = kRowsAhead / 2;        // rows ahead per chain (two chains at even T)
    if (T == 1) {
#pragma unroll
      for (int k = 0; k < NC; ++k) putk(k, 0, zk(k, 0));
      return;
    }
    float even[NC], cur[NC];
    if ((T & 1) == 0) {
      // even T: two independent chains (even rows upwards from row 0, odd rows downwards from row T-1), walked together
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        even[k] = zk(k, 0);
        cur[k] = zk(k, T - 1);
        putk(k, 0, even[k]);
      }
      const long long J = T / 2 - 1;
      long long j = 0;
      for (; j + RA <= J; j += RA) {
        float ve[NC][RA], vo[NC][RA];
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
          for (int r = 0; r < RA; ++r) {
            const long long je = 2 * (j + r + 1), io = T - 1 - 2 * (j + r);
            ve[k][r] = src_row0[(je - 1) * ldy + fk[k]];
            vo[k][r] = src_row0[(io - 1) * ldy + fk[k]];
          }
#pragma unroll
        for (int r = 0; r < RA; ++r)
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            const long long je = 2 * (j + r + 1), io = T - 1 - 2 * (j + r);
            even[k] = even[k] + 4.0f * prep(je - 1, ve[k][r]);
            putk(k, je, even[k]);
            cur[k] = cur[k] - 4.0f * prep(io - 1, vo[k][r]);
            putk(k, io - 2, cur[k]);
          }
        __syncthreads();
      }
      for (; j < J; ++j)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const long long je = 2 * (j + 1), io = T - 1 - 2 * j;
          even[k] = even[k] + 4.0f * zk(k, je - 1);
          putk(k, je, even[k]);
          cur[k] = cur[k] - 4.0f * zk(k, io - 1);
          putk(k, io - 2, cur[k]);
        }
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        cur[k] = cur[k] - 4.0f * zk(k, 0);      // i = 1: out[-1]
        putk(k, T - 1, cur[k]);
      }
      return;
    }
    // odd T: the second loop starts from the first chain's LAST value and rewrites every even row the first one wrote
    // (out[i-2] = out[i] - 4 x[i-1], i = T-1 ... 2), the odd rows stay 0.  So the first chain is walked for its last value
    // only -- its stores would all be overwritten -- and every row is written once.
#pragma unroll
    for (int k = 0; k < NC; ++k) even[k] = zk(k, 0);
    {
      long long i = 2;
      for (; i + 2 * (kRowsAhead - 1) < T; i += 2 * kRowsAhead) {
        float v[NC][kRowsAhead];
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
          for (int r = 0; r < kRowsAhead; ++r) v[k][r] = src_row0[(i + 2 * r - 1) * ldy + fk[k]];
#pragma unroll
        for (int r = 0; r < kRowsAhead; ++r)
#pragma unroll
          for (int k = 0; k < NC; ++k) even[k] = even[k] + 4.0f * prep(i + 2 * r - 1, v[k][r]);
        __syncthreads();
      }
      for (; i < T; i += 2)
#pragma unroll
        for (int k = 0; k < NC; ++k) even[k] = even[k] + 4.0f * zk(k, i - 1);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      cur[k] = even[k];
      putk(k, T - 1, cur[k]);
    }
    {
      long long i = T - 1;
      for (; i - 2 * (kRowsAhead - 1) >= 2; i -= 2 * kRowsAhead) {
        float v[NC][kRowsAhead];
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
          for (int r = 0; r < kRowsAhead; ++r) v[k][r] = src_row0[(i - 2 * r - 1) * ldy + fk[k]];
#pragma unroll
        for (int r = 0; r < kRowsAhead; ++r)
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            const long long ii = i - 2 * r;
            cur[k] = cur[k] - 4.0f * prep(ii - 1, v[k][r]);
            putk(k, ii - 2, cur[k]);
            putk(k, ii - 1, 0.0f);
          }
        __syncthreads();
      }
      for (; i >= 2; i -= 2)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          cur[k] = cur[k] - 4.0f * zk(k, i - 1);
          putk(k, i - 2, cur[k]);
          putk(k, i - 1, 0.0f);
        }
    }
  } else {
    // fint_central (utils/misc.py:96-104), statement by statement.  Rows the reference never writes stay 0.
    auto z = [&](long long t) { return prep(t, src[t * ldy]); };
    auto put = [&](long long row, float ph) { put_phase<POLAR>(p, base + row * F, ph, mag_at<POLAR>(p, base + row * F)); };
    if (T == 1) {
      put(0, z(0));
      return;
    }
    if ((T & 1) == 0) {
      // Even T: the first loop writes the even rows, the second starts from x[T-1] and writes the odd ones (row "-1" =
      // T-1 last): two independent chains, walked together -- twice the rows in flight, every row written once.
      float even = z(0), cur = z(T - 1);
      put(0, even);
      const long long J = T / 2 - 1;          // paired steps; the second chain has one more (i = 1)
      auto pair = [&](long long j, float ve, float vo, float me, float mo) {
        const long long je = 2 * (j + 1), io = T - 1 - 2 * j;
        even = even + 4.0f * prep(je - 1, ve);
        put_phase<POLAR>(p, base + je * F, even, me);
        cur = cur - 4.0f * prep(io - 1, vo);
        put_phase<POLAR>(p, base + (io - 2) * F, cur, mo);
      };
      long long j = 0;
      for (; j + kRowsAhead <= J; j += kRowsAhead) {
        float ve[kRowsAhead], vo[kRowsAhead], me[kRowsAhead], mo[kRowsAhead];
#pragma unroll
        for (int k = 0; k < kRowsAhead; ++k) {
          const long long je = 2 * (j + k + 1), io = T - 1 - 2 * (j + k);
          ve[k] = src[(je - 1) * ldy];
          vo[k] = src[(io - 1) * ldy];
          me[k] = mag_at<POLAR>(p, base + je * F);
          mo[k] = mag_at<POLAR>(p, base + (io - 2) * F);
        }
#pragma unroll
        for (int k = 0; k < kRowsAhead; ++k) pair(j + k, ve[k], vo[k], me[k], mo[k]);
      }
      for (; j < J; ++j) {
        const long long je = 2 * (j + 1), io = T - 1 - 2 * j;
        pair(j, src[(je - 1) * ldy], src[(io - 1) * ldy], mag_at<POLAR>(p, base + je * F), mag_at<POLAR>(p, base + (io - 2) * F));
      }
      cur = cur - 4.0f * z(0);                // i = 1: out[-1]
      put(T - 1, cur);
      return;
    }
    // Odd T (even T returned above): the second loop starts from the first chain's LAST value and rewrites every even
    // row the first one wrote (out[i-2] = out[i] - 4 x[i-1], i = T-1 ... 2); the odd rows stay 0.  The first chain is
    // therefore walked for its last value only (its stores were all dead) and every row is written once, odd rows with
    // their even neighbours.
    float even = z(0);
    long long i = 2;
    for (; i + 2 * (kRowsAhead - 1) < T; i += 2 * kRowsAhead) {   // out[i] = out[i-2] + 4 x[i-1]
      float v[kRowsAhead];
#pragma unroll
      for (int k = 0; k < kRowsAhead; ++k) v[k] = src[(i + 2 * k - 1) * ldy];
#pragma unroll
      for (int k = 0; k < kRowsAhead; ++k) even = even + 4.0f * prep(i + 2 * k - 1, v[k]);
    }
    for (; i < T; i += 2) even = even + 4.0f * z(i - 1);
    float cur = even;
    put(T - 1, cur);
    i = T - 1;
    for (; i - 2 * (kRowsAhead - 1) >= 2; i -= 2 * kRowsAhead) {   // out[i-2] = out[i] - 4 x[i-1]
      float v[kRowsAhead], m[kRowsAhead], mz[kRowsAhead];
#pragma unroll
      for (int k = 0; k < kRowsAhead; ++k) {
        const long long ii = i - 2 * k;
        v[k] = src[(ii - 1) * ldy];
        m[k] = mag_at<POLAR>(p, base + (ii - 2) * F);
        mz[k] = mag_at<POLAR>(p, base + (ii - 1) * F);
      }
#pragma unroll
      for (int k = 0; k < kRowsAhead; ++k) {
        const long long ii = i - 2 * k;
        cur = cur - 4.0f * prep(ii - 1, v[k]);
        put_phase<POLAR>(p, base + (ii - 2) * F, cur, m[k]);
        put_phase<POLAR>(p, base + (ii - 1) * F, 0.0f, mz[k]);
      }
    }
    for (; i >= 2; i -= 2) {
      cur = cur - 4.0f * z(i - 1);
      put(i - 2, cur);
      put(i - 1, 0.0f);
    }
  }
}

// mag * exp(i phase) (SpectralRepresentation.invert, spectral_repr.py:449-451)
__global__ __launch_bounds__(256) void polar_to_complex_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                                long long n, float2* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s, c;
    fast_sincosf(phase[i], s, c);
    const float m = mag[i];
    out[i] = make_float2(m * c, m * s);
  }
}

// run-time options -> template flags
// columns per thread of the clip-per-block layout (0: flattened columns): rows that are not a whole number of 64-byte
// segments, long enough for a block of their own, short enough for 1024 threads x 4
static int clip_block_columns(long long F, long long B, int elem_bytes) {
  if (variant(kVarScanLayout) == 1) return 0;
  if ((F * elem_bytes) % 64 == 0 || F < 256 || B < 64) return 0;
  return F <= 2048 ? 2 : F <= 4096 ? 4 : 0;
}
static unsigned clip_block_threads(long long F, int nc) { return (unsigned)((((F + nc - 1) / nc) + 63) / 64 * 64); }

template <int MODE, bool CPLX, bool WIN, bool NORM>
static void launch_scan4(dim3 grid, dim3 block, hipStream_t s, const ScanParams& p) {
  const int nc = clip_block_columns(p.F, p.B, 4);
  if (nc == 2) hipLaunchKernelGGL((phase_scan_kernel<MODE, CPLX, WIN, NORM, 2>), dim3((unsigned)p.B), dim3(clip_block_threads(p.F, 2)), 0, s, p);
  else if (nc == 4) hipLaunchKernelGGL((phase_scan_kernel<MODE, CPLX, WIN, NORM, 4>), dim3((unsigned)p.B), dim3(clip_block_threads(p.F, 4)), 0, s, p);
  else hipLaunchKernelGGL((phase_scan_kernel<MODE, CPLX, WIN, NORM, 1>), grid, block, 0, s, p);
}
template <int MODE, bool CPLX, bool WIN>
static void launch_scan3(bool norm, dim3 grid, dim3 block, hipStream_t s, const ScanParams& p) {
  if (norm) launch_scan4<MODE, CPLX, WIN, true>(grid, block, s, p);
  else launch_scan4<MODE, CPLX, WIN, false>(grid, block, s, p);
}
template <int MODE>
static void launch_scan1(bool cplx, bool win, bool norm, dim3 grid, dim3 block, hipStream_t s, const ScanParams& p) {
  if (cplx) {
    if (win) launch_scan3<MODE, true, true>(norm, grid, block, s, p);
    else launch_scan3<MODE, true, false>(norm, grid, block, s, p);
  } else {
    if (win) launch_scan3<MODE, false, true>(norm, grid, block, s, p);
    else launch_scan3<MODE, false, false>(norm, grid, block, s, p);
  }
}
static void launch_scan(int mode, bool cplx, bool win, bool norm, dim3 grid, dim3 block, hipStream_t s, const ScanParams& p) {
  switch (mode) {
    case SCAN_UNWRAP: launch_scan1<SCAN_UNWRAP>(cplx, win, norm, grid, block, s, p); break;
    case SCAN_IF_FORWARD: launch_scan1<SCAN_IF_FORWARD>(cplx, win, norm, grid, block, s, p); break;
    case SCAN_IF_BACKWARD: launch_scan1<SCAN_IF_BACKWARD>(cplx, win, norm, grid, block, s, p); break;
    case SCAN_IF_CENTRAL: launch_scan1<SCAN_IF_CENTRAL>(cplx, win, norm, grid, block, s, p); break;
    default: launch_scan1<SCAN_ANGLE>(cplx, win, norm, grid, block, s, p); break;
  }
}

}  // namespace at_hip

using namespace at_hip;

static int phase_scan_impl(const float* X_complex, const float* phase, int64_t B, int64_t T, int64_t F, int mode, int bare,
                           const float* frame_window, const float* offset, const float* scale, float* out, int64_t ld_out,
                           void* stream) {
  if (B < 0 || T < 0 || F < 0 || ld_out < F) return AT_EINVAL;
  if (B * T * F == 0) return AT_OK;
  if ((X_complex == nullptr) == (phase == nullptr) || !out) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (mode < SCAN_UNWRAP || mode > SCAN_ANGLE) return AT_EINVAL;
  if (bare && (mode < SCAN_IF_FORWARD || mode > SCAN_IF_CENTRAL || !phase)) return AT_EINVAL;
  ScanParams p = {(const float2*)X_complex, phase, out, B, T, F, frame_window, offset, scale, bare, ld_out};
  if (mode == SCAN_ANGLE) {
    long long blocks = (B * T * F + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (X_complex) hipLaunchKernelGGL(phase_angle_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(phase_angle_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
  }
  const long long cols = B * F;
  const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
  launch_scan(mode, X_complex != nullptr, frame_window != nullptr, offset != nullptr, grid, block, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

// PolarIF.forward in one pass (clip-per-block layout only): IF scan + banded magnitude of the same rows
template <int MODE, bool WIN, bool NORM>
static void launch_scan_mag(int nc, size_t lds, hipStream_t s, const ScanParams& p) {
  (void)nc;   // two columns per thread only: four need more registers than a 1024-thread block has (spills)
  hipLaunchKernelGGL((phase_scan_kernel<MODE, true, WIN, NORM, 2, true>), dim3((unsigned)p.B), dim3(clip_block_threads(p.F, 2)), lds, s, p);
}
template <int MODE>
static void launch_scan_mag1(bool win, bool norm, int nc, size_t lds, hipStream_t s, const ScanParams& p) {
  if (win) {
    if (norm) launch_scan_mag<MODE, true, true>(nc, lds, s, p);
    else launch_scan_mag<MODE, true, false>(nc, lds, s, p);
  } else {
    if (norm) launch_scan_mag<MODE, false, true>(nc, lds, s, p);
    else launch_scan_mag<MODE, false, false>(nc, lds, s, p);
  }
}

template <bool POLAR>
static int phase_integrate_impl(const float* y, int64_t ld_y, int64_t B, int64_t T, int64_t F, int method, int rescale,
                                const float* offset, const float* scale, const float* mag, float* out, void* stream) {
  if (B < 0 || T < 0 || F < 0 || ld_y < F) return AT_EINVAL;
  if (B * T * F == 0) return AT_OK;
  if (!y || !out || y == out || (POLAR && !mag)) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (method < SCAN_IF_FORWARD || method > SCAN_IF_CENTRAL) return AT_EINVAL;
  IntParams p = {y, out, B, T, F, offset, scale, ld_y, mag};
  const long long cols = B * F;
  const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool norm = offset != nullptr;
  void (*kernel)(IntParams, int) = nullptr;
  // the complex-output form (mag * exp(i phase): sincos and a second input per element) is slower in the clip-per-block
  // layout (PolarIF.invert 2.30 -> 3.29 ms): it stays on flattened columns
  const int nc = POLAR ? 0 : clip_block_columns(F, B, 4);
  dim3 g = grid, blk = block;
  if (nc) {
    g = dim3((unsigned)B);
    blk = dim3(clip_block_threads(F, nc));
  }
  if (method == SCAN_IF_FORWARD && nc == 2)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_FORWARD, true, POLAR, 2> : phase_integrate_kernel<SCAN_IF_FORWARD, false, POLAR, 2>;
  else if (method == SCAN_IF_FORWARD && nc == 4)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_FORWARD, true, POLAR, 4> : phase_integrate_kernel<SCAN_IF_FORWARD, false, POLAR, 4>;
  else if (method == SCAN_IF_BACKWARD && nc == 2)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_BACKWARD, true, POLAR, 2> : phase_integrate_kernel<SCAN_IF_BACKWARD, false, POLAR, 2>;
  else if (method == SCAN_IF_BACKWARD && nc == 4)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_BACKWARD, true, POLAR, 4> : phase_integrate_kernel<SCAN_IF_BACKWARD, false, POLAR, 4>;
  else if (method == SCAN_IF_CENTRAL && nc == 2)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_CENTRAL, true, false, 2> : phase_integrate_kernel<SCAN_IF_CENTRAL, false, false, 2>;
  else if (method == SCAN_IF_CENTRAL && nc == 4)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_CENTRAL, true, false, 4> : phase_integrate_kernel<SCAN_IF_CENTRAL, false, false, 4>;
  else if (method == SCAN_IF_FORWARD)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_FORWARD, true, POLAR, 1> : phase_integrate_kernel<SCAN_IF_FORWARD, false, POLAR, 1>;
  else if (method == SCAN_IF_BACKWARD)
    kernel = norm ? phase_integrate_kernel<SCAN_IF_BACKWARD, true, POLAR, 1> : phase_integrate_kernel<SCAN_IF_BACKWARD, false, POLAR, 1>;
  else
    kernel = norm ? phase_integrate_kernel<SCAN_IF_CENTRAL, true, POLAR, 1> : phase_integrate_kernel<SCAN_IF_CENTRAL, false, POLAR, 1>;
  hipLaunchKernelGGL(kernel, g, blk, 0, s, p, rescale);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

extern "C" {

int at_phase_scan(const float* X_complex, const float* phase, int64_t B, int64_t T, int64_t F, int mode, int bare,
                  const float* frame_window, const float* offset, const float* scale, float* out, void* stream) {
  return phase_scan_impl(X_complex, phase, B, T, F, mode, bare, frame_window, offset, scale, out, F, stream);
}

int at_phase_scan_strided(const float* X_complex, const float* phase, int64_t B, int64_t T, int64_t F, int mode, int bare,
                          const float* frame_window, const float* offset, const float* scale, float* out, int64_t ld_out,
                          void* stream) {
  return phase_scan_impl(X_complex, phase, B, T, F, mode, bare, frame_window, offset, scale, out, ld_out, stream);
}

int at_polarif_forward(const float* X_complex, int64_t B, int64_t T, int64_t F, int method, const float* frame_window,
                       const float* if_offset, const float* if_scale, const int* band_start, const int* band_len,
                       const int* band_off, const float* band_w, int64_t n_w, int contrast, const float* mag_offset,
                       const float* mag_scale, float eps, float* out_stacked, void* stream) {
  if (B < 0 || T < 0 || F < 0 || n_w < 0) return AT_EINVAL;
  if (B * T * F == 0) return AT_OK;
  if (!X_complex || !out_stacked || !band_start || !band_len || !band_off || (n_w > 0 && !band_w)) return AT_EINVAL;
  if ((if_offset == nullptr) != (if_scale == nullptr) || (mag_offset == nullptr) != (mag_scale == nullptr)) return AT_EINVAL;
  if (method < SCAN_IF_FORWARD || method > SCAN_IF_CENTRAL || contrast < C_NONE || contrast > C_LOG10) return AT_EINVAL;
  const int nc = clip_block_columns(F, B, 4);
  const int row_floats = (int)((F + 3) & ~3ll) + 4;
  const size_t lds = sizeof(float) * (size_t)(((n_w + 3) & ~3ll) + (long long)kRowsAhead * row_floats);
  if (nc != 2 || lds > 64 * 1024) return AT_EUNSUPPORTED;      // the caller runs the two stand-alone kernels instead
  ScanParams p = {(const float2*)X_complex, nullptr, out_stacked + F, B, T, F, frame_window, if_offset, if_scale, 0, 2 * F};
  p.mag_out = out_stacked;
  p.band_start = band_start;
  p.band_len = band_len;
  p.band_off = band_off;
  p.band_w = band_w;
  p.n_w = (int)n_w;
  p.row_floats = row_floats;
  p.contrast = contrast;
  p.eps = eps;
  p.mag_offset = mag_offset;
  p.mag_scale = mag_scale;
  hipStream_t s = (hipStream_t)stream;
  const bool win = frame_window != nullptr, norm = if_offset != nullptr;
  if (method == SCAN_IF_FORWARD) launch_scan_mag1<SCAN_IF_FORWARD>(win, norm, nc, lds, s, p);
  else if (method == SCAN_IF_BACKWARD) launch_scan_mag1<SCAN_IF_BACKWARD>(win, norm, nc, lds, s, p);
  else launch_scan_mag1<SCAN_IF_CENTRAL>(win, norm, nc, lds, s, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_phase_integrate(const float* y, int64_t B, int64_t T, int64_t F, int method, int rescale, const float* offset,
                       const float* scale, float* out, void* stream) {
  return phase_integrate_impl<false>(y, F, B, T, F, method, rescale, offset, scale, nullptr, out, stream);
}

int at_phase_integrate_polar(const float* y, int64_t ld_y, int64_t B, int64_t T, int64_t F, int method, const float* offset,
                             const float* scale, const float* mag, float* out_complex, void* stream) {
  return phase_integrate_impl<true>(y, ld_y, B, T, F, method, 1, offset, scale, mag, out_complex, stream);
}

int at_polar_to_complex(const float* mag, const float* phase, int64_t n, float* out_complex, void* stream) {
  if (n < 0) return AT_EINVAL;
  if (n == 0) return AT_OK;
  if (!mag || !phase || !out_complex) return AT_EINVAL;
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(polar_to_complex_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mag, phase,
                     (long long)n, (float2*)out_complex);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"
