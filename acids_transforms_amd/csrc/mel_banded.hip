// mel_banded.hip -- stand-alone projection of a stored spectrum onto a *banded* filterbank, with the same
// prologue / epilogue options as the dense MFMA projection of mel.hip:
//     forward  normalise(contrast(|x|^p @ bank))                      spectral_repr.py:215-226, mel.py:68-73
//     inverse  invert_contrast(y * scale + offset) @ inverse_bank      spectral_repr.py:228-240
// Mel banks -- the reference's default 513 x 513 one included, and its row-normalised transpose used by
// `invert` -- are banded: column n is non-zero on a short run of rows only.  A dense contraction spends
// 2 K N flops per frame on what are ~1000 useful multiply-adds; it is MFMA-bound at 5 ms for 1024 clips where
// the data could stream through in under 1 ms.  Here one wavefront takes one frame at a time: the K input values
// (K <= 2112: every n_fft up to 4096) go through the prologue into an LDS row, then every lane walks the band of
// one filter per pass exactly as the fused STFT epilogue does (band_bank.h, utils/banded.py: conflict-free
// ds_read_b128 of values and weights), and the epilogue writes N outputs.  HBM-bound: the input row in, N floats
// out.  The weight table, the lane tables and one row per wave share the 160 KB of LDS; the launcher drops from
// eight to four waves per workgroup when a long row and a large table would not fit together.
#include <hip/hip_runtime.h>
#include "fastmath.h"
#include <stdint.h>
#include <type_traits>

#include "../../include/acids_hip.h"
#include "band_bank.h"
#include "run_plan.h"
#include <stdlib.h>
#include <string.h>
#include "mel_gemm.h"   // A_* / C_* codes
#include "variants.h"

namespace at_hip {

struct BandedParams {
  const void* A;        // rows x K (complex64 or float32), contiguous rows of lda elements
  float* out;
  const float* offset;  // device scalars (may be null)
  const float* scale;
  long long rows, lda, ld_out;
  long long T;          // > 0: channel-major store out[(r / T) * N * T + n * T + r % T]
  long long rows_per_wave;
  BandBank bank;
  int K, a_kind, contrast, inverse;
  int row_floats;       // LDS row: the K values + zero padding a walk may run into (multiple of 64)
  int table_floats;     // 64 * sum(pass_len)
  float eps;
  // optional second output for complex input: normalise(angle(x)), rows of ld_phase floats (Polar: the stacked
  // (.., T, 2, F) tensor is written in place -- magnitudes at row offset 0, phases at row offset F)
  float* phase_out;
  long long ld_phase;
  const float* ph_offset;
  const float* ph_scale;
  // inverse only: phase_in != null turns the output into complex64, out[r, f] = acc * exp(i * denormalise(
  // phase_in[r * ld_phase + f])) -- Polar.invert (spectral_repr.py:441-451) in one pass over the stacked tensor
  const float* phase_in;
};

constexpr int kBandedWaves = 8;     // waves per workgroup (at most), sharing the LDS weight table
constexpr int kMaxRowK = 2112;      // longest input row: 33 segments of 64 (n_fft 4096 -> 2049 bins)
constexpr int kMaxWalk = 512;       // longest band a lane walks
constexpr size_t kLdsBudget = 160 * 1024;

// sum over one band: `quads` steps of four bins, values at a[j], this lane's weights at w[64 j].  The reads of four steps
// are issued before the first multiply-add (one LDS round trip per four steps instead of one per step); the chain of
// multiply-adds keeps the order of the plain loop, so the result does not change.
__device__ __forceinline__ float band_dot(const float4* a, const float4* w, int quads) {
  float acc = 0.f;
  int j = 0;
  for (; j + 4 <= quads; j += 4) {
    float4 av[4], wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = a[j + u];
      wv[u] = w[(j + u) * 64];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = fmaf(av[u].x, wv[u].x, acc);
      acc = fmaf(av[u].y, wv[u].y, acc);
      acc = fmaf(av[u].z, wv[u].z, acc);
      acc = fmaf(av[u].w, wv[u].w, acc);
    }
  }
  for (; j < quads; ++j) {
    const float4 av = a[j], wv = w[j * 64];
    acc = fmaf(av.x, wv.x, acc);
    acc = fmaf(av.y, wv.y, acc);
    acc = fmaf(av.z, wv.z, acc);
    acc = fmaf(av.w, wv.w, acc);
  }
  return acc;
}

// NSEG >= ceil(K / 64) segments of 64 values per row; EXACT: NSEG == ceil(K / 64) (only the last segment can run
// past the row's end).  CMW = 1 / 2: channel-major output (p.T > 0) of a bank with that many passes through a register
// window -- every lane keeps the last eight frames of its filter(s) and stores them as 32 contiguous bytes of the
// (.., N, T) tensor (a 4-byte store per frame leaves partly written lines to be fetched again: MelSpectrogram / MFCC at
// n_fft 2048, 0.90 -> 0.6 ms per 1024 clips).

template <bool CPLX, int NSEG, bool EXACT, int CMW = 0>
__global__ __launch_bounds__(64 * kBandedWaves) void mel_banded_kernel(BandedParams p) {
  // passes whose phase inputs are requested with the row (Polar.invert): a bank of K filters over K bins has as many passes
  // as the row has segments -- 9 at n_fft 1024, 17 at 2048; the 33 of n_fft 4096 keep the load inside the pass (requested
  // ahead they change nothing there: 2.73 ms either way)
  constexpr int kPhaseAhead = NSEG <= 17 ? NSEG : 1;
  extern __shared__ float4 band_lds[];   // weight table, lane_start / lane_filter, one row per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row bookkeeping on the scalar unit
  const int n_thr = blockDim.x;
  float* wlds = reinterpret_cast<float*>(band_lds);
  const int table_floats = p.table_floats;
  for (int i = threadIdx.x; i < table_floats; i += n_thr) wlds[i] = p.bank.weights[i];
  int* lane_tab = reinterpret_cast<int*>(wlds + table_floats);
  for (int i = threadIdx.x; i < 64 * p.bank.n_passes; i += n_thr) {
    lane_tab[i] = p.bank.lane_start[i];
    lane_tab[64 * p.bank.n_passes + i] = p.bank.lane_filter[i];
  }
  float* absrow = reinterpret_cast<float*>(lane_tab + 2 * 64 * p.bank.n_passes) + wave * p.row_floats;
  for (int k = 64 * NSEG + lane; k < p.row_floats; k += 64) absrow[k] = 0.0f;   // padding behind the row stays zero
  __syncthreads();

  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }
  float ph_off = 0.f, ph_sc = 1.f;
  if (p.ph_offset) {
    ph_off = *p.ph_offset;
    ph_sc = *p.ph_scale;
  }
  // Normalize.forward is (x - offset) / scale (norm.py:40-41); as in the fused epilogues the quotient is a multiplication
  // by the reciprocal taken once per wave (<= 1.5 ulp from the exact division, against the 1e-5 bar)
#ifdef AT_BANDED_EXACT_DIV
#define AT_NORM(v, o, s, inv) (((v) - (o)) / (s))
#else
#define AT_NORM(v, o, s, inv) (((v) - (o)) * (inv))
#endif
  const float inv_sc = 1.0f / sc, ph_inv = 1.0f / ph_sc;
  (void)inv_sc;
  (void)ph_inv;
  const long long w_id = (long long)blockIdx.x * (n_thr >> 6) + wave;
  long long r = w_id * p.rows_per_wave;
  long long r_end = r + p.rows_per_wave;
  if (r_end > p.rows) r_end = p.rows;
  if (r >= r_end) return;

  using In = typename std::conditional<CPLX, float2, float>::type;
  const In* A = reinterpret_cast<const In*>(p.A);
  // channel-major register window: (clip, frame) of the current row tracked incrementally
  float cm[CMW > 0 ? CMW : 1][8];
#pragma unroll
  for (int q = 0; q < (CMW > 0 ? CMW : 1); ++q)
#pragma unroll
    for (int k = 0; k < 8; ++k) cm[q][k] = 0.f;
  long long cb = 0, ct = 0;
  long long e_next[CMW > 0 ? CMW : 1];
  bool e_valid = false;
  int held[CMW > 0 ? CMW : 1];                    // frames in the window that are not stored yet (<= 8)
#pragma unroll
  for (int q = 0; q < (CMW > 0 ? CMW : 1); ++q) {
    held[q] = 0;
    e_next[q] = 0;
  }
  if (CMW > 0) {
    cb = r / p.T;
    ct = r - cb * p.T;
  }
  // Unconditional loads only (the last segment re-reads element K-1 on the lanes past the row's end; `walk` zeroes
  // them): behind a conditional load the compiler drains vmcnt on the spot, and the next row, meant to arrive while
  // this one is walked, was then waited for before the walk began.
  auto fetch = [&](long long row, In (&v)[NSEG]) {
    const In* src = A + row * p.lda;
#pragma unroll
    for (int m = 0; m < NSEG; ++m) {
      int k = lane + 64 * m;
      if (!EXACT || m + 1 == NSEG) k = k < p.K ? k : p.K - 1;
      v[m] = src[k];
    }
  };
  auto walk = [&](long long r, const In (&cur)[NSEG], auto phase_ahead) {   // phase_ahead: float[kPhaseAhead] or nullptr
    // prologue into the LDS row
#pragma unroll
    for (int m = 0; m < NSEG; ++m) {
      float v;
      if constexpr (CPLX) {
        const float s2 = fmaf(cur[m].x, cur[m].x, cur[m].y * cur[m].y);
        v = (p.a_kind == A_COMPLEX_ABS2) ? s2 : __builtin_amdgcn_sqrtf(s2);
        if (p.phase_out) {
          const int kk = lane + 64 * m;
          if ((EXACT && m + 1 < NSEG) || kk < p.K) {
            float ph = fast_atan2f(cur[m].y, cur[m].x);
            if (p.ph_offset) ph = AT_NORM(ph, ph_off, ph_sc, ph_inv);
            p.phase_out[r * p.ld_phase + kk] = ph;
          }
        }
      } else {
        v = cur[m];
        if (p.a_kind == A_REAL_ABS) v = fabsf(v);
        if (p.inverse) {
          if (p.offset) v = __fadd_rn(__fmul_rn(v, sc), off);
          v = banded_contrast_inv(v, p.contrast, p.eps);
        }
      }
      const int k = lane + 64 * m;
      absrow[k] = ((EXACT && m + 1 < NSEG) || k < p.K) ? v : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float4* w = reinterpret_cast<const float4*>(wlds) + lane;
    if constexpr (CMW > 0) {
      int fq[CMW];
#pragma unroll
      for (int q = 0; q < CMW; ++q) {
        fq[q] = lane_tab[(CMW + q) * 64 + lane];
        const float4* a = reinterpret_cast<const float4*>(absrow + lane_tab[q * 64 + lane]);
        float acc = 0.f;
        const int quads = p.bank.pass_len[q] >> 2;
        acc = band_dot(a, w, quads);
        w += quads * 64;
        acc = banded_contrast_fwd(acc, p.contrast, p.eps);
        if (p.offset) acc = AT_NORM(acc, off, sc, inv_sc);
#pragma unroll
        for (int k = 0; k < 7; ++k) cm[q][k] = cm[q][k + 1];
        cm[q][7] = acc;
      }
      const long long b = cb, t = ct;
      if (++ct == p.T) {
        ct = 0;
        ++cb;
      }
      // Every lane flushes its window when the eight frames it holds end on a 32-byte boundary of its own output
      // row ((.., N, T) rows start at arbitrary multiples of 4 bytes: T is odd as often as not), so that a store
      // covers whole 32-byte sectors -- and at the end of a clip or of this wave's rows, whatever it holds.
      const bool last_of_run = (t == p.T - 1) || (r == r_end - 1);
#pragma unroll
      for (int q = 0; q < CMW; ++q) {
        ++held[q];
        if (fq[q] >= 0) {
          // element index one past frame t in this lane's output row: (b N + f) T + t + 1, tracked incrementally
          if (!e_valid) e_next[q] = (b * p.bank.n_filters + fq[q]) * p.T + t + 1;
          const long long e = e_next[q];
          e_next[q] = e + ((t == p.T - 1) ? (long long)(p.bank.n_filters - 1) * p.T + 1 : 1);
          if ((e & 7) == 0 || last_of_run) {
            float* dst = p.out + e - 8;
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
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      return;
    }
    auto pass = [&](int q, float ph_ahead, bool have_phase) {
      const int f = lane_tab[(p.bank.n_passes + q) * 64 + lane];
      const float4* a = reinterpret_cast<const float4*>(absrow + lane_tab[q * 64 + lane]);
      float acc = 0.f;
      const int quads = p.bank.pass_len[q] >> 2;
      acc = band_dot(a, w, quads);
      w += quads * 64;
      if (f >= 0) {
        if (!p.inverse) {
          acc = banded_contrast_fwd(acc, p.contrast, p.eps);
          if (p.offset) acc = AT_NORM(acc, off, sc, inv_sc);
        }
        if (p.phase_in) {
          float ph = have_phase ? ph_ahead : p.phase_in[r * p.ld_phase + f];
          if (p.ph_offset) ph = __fadd_rn(__fmul_rn(ph, ph_sc), ph_off);
          float sn, cs;
          fast_sincosf(ph, sn, cs);
          reinterpret_cast<float2*>(p.out)[r * p.ld_out + f] = make_float2(acc * cs, acc * sn);
        } else if (p.T > 0) {
          const long long b = r / p.T, t = r - b * p.T;
          p.out[(b * p.bank.n_filters + f) * p.T + t] = acc;
        } else {
          p.out[r * p.ld_out + f] = acc;
        }
      }
    };
    if constexpr (std::is_same<decltype(phase_ahead), std::nullptr_t>::value) {
      for (int q = 0; q < p.bank.n_passes; ++q) pass(q, 0.f, false);
    } else {
#pragma unroll
      for (int q = 0; q < kPhaseAhead; ++q)
        if (q < p.bank.n_passes) pass(q, phase_ahead[q], true);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // two register sets swapping roles (a copy of registers with loads in flight would wait for them -- and, memory
  // returning in order, for this row's stores); the last row of a run requests itself again
  In ra[NSEG];
  if constexpr (!CPLX || NSEG > 17) {
    // real rows (the inverse projection, |x| inputs): half the bytes per row and exp() per element -- the second
    // copy of the loop body costs more than the prefetch returns (A/B: 0.73 vs 0.70 ms); occupancy hides the load.
    // The longest complex rows (33 segments: 66 registers a copy) take the same single-buffered loop.
    if constexpr (!CPLX && CMW == 0) {
      // Polar.invert: the phases of a row are requested with the row, not one per pass behind the previous pass's
      // store (loads and stores share vmcnt and return in order: every pass waited for a store's round trip)
      if (p.phase_in && p.bank.n_passes <= kPhaseAhead) {
        // the column of every pass in registers (read back from the LDS table at every row instead: 1.42 -> 1.66 ms)
        int fcol[kPhaseAhead];
#pragma unroll
        for (int q = 0; q < kPhaseAhead; ++q) {
          const int f = q < p.bank.n_passes ? lane_tab[(p.bank.n_passes + q) * 64 + lane] : 0;
          fcol[q] = f > 0 ? f : 0;
        }
        auto fetch_phases = [&](long long row, float (&ph)[kPhaseAhead]) {
          const float* prow = p.phase_in + row * p.ld_phase;
#pragma unroll
          for (int q = 0; q < kPhaseAhead; ++q) ph[q] = prow[fcol[q]];
        };
        // (one row ahead on two register sets, as the complex rows do it, measured slower here: 1.53 against 1.38 ms)
        float pa[kPhaseAhead];
        for (; r < r_end; ++r) {
          fetch(r, ra);
          fetch_phases(r, pa);
          walk(r, ra, pa);
        }
        return;
      }
    }
    for (; r < r_end; ++r) {
      fetch(r, ra);
      walk(r, ra, nullptr);
    }
    return;
  } else {
  In rb[NSEG];
  fetch(r, ra);
  while (r < r_end) {
    fetch(r + 1 < r_end ? r + 1 : r, rb);
    walk(r, ra, nullptr);
    if (++r >= r_end) break;
    fetch(r + 1 < r_end ? r + 1 : r, ra);
    walk(r, rb, nullptr);
    ++r;
  }
  }
}


// ---------------------------------------------------------------------------
// Fixed form of the headline projection (round 3): complex rows of 513 bins -> normalise(log1p(|X| @ bank)) for a
// two-pass bank whose walks are FQ0 and FQ1 quads long (128 mel filters at 44.1 kHz: 8 and 2), row-major output.
// Everything the general kernel above decides at run time (input kind, contrast, direction, phase side outputs,
// layout, pass count and lengths) is fixed here, and the arithmetic is the fused n_fft-1024 epilogue's own
// (band_walk_fixed, hardware log2, reciprocal of the scale): `Magnitude.forward` on a stored spectrum and the fused
// `STFT + Magnitude` kernel give the same bits.  One wave per row, rows requested one ahead (two register sets swapping
// roles), eight waves per block sharing the 10 KB weight table.
// ---------------------------------------------------------------------------
struct FixedProjParams {
  const float2* X;       // rows x 513 complex64, contiguous
  float* out;            // rows x N
  const float* offset;   // device scalars or null
  const float* scale;
  const float* weights;  // band table, 64 * 4 * (FQ0 + FQ1) floats
  const int* lane_start; // [2][64]
  const int* lane_filter;
  long long rows, rows_per_wave;
  int N;
};

template <int FQ0, int FQ1>
__global__ __launch_bounds__(64 * kBandedWaves, 4) void mel_fixed_kernel(FixedProjParams p) {
  constexpr int kTable = 64 * 4 * (FQ0 + FQ1);
  constexpr int kRow = 640;                       // 513 bins + what a walk may run past (zero weights there)
  __shared__ __attribute__((aligned(16))) float wlds[kTable];
  __shared__ __attribute__((aligned(16))) float rows_lds[kBandedWaves][kRow];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < kTable; i += 64 * kBandedWaves) wlds[i] = p.weights[i];
  float* absrow = rows_lds[wave];
  absrow[576 + lane] = 0.0f;                      // 513 .. 575 are rewritten per row (zeros), 576 .. 639 stay zero
  __syncthreads();
  const int start0 = p.lane_start[lane], start1 = p.lane_start[64 + lane];
  const int f0 = p.lane_filter[lane], f1 = p.lane_filter[64 + lane];
  float off = 0.f, inv = 1.f;
  if (p.offset) {
    off = *p.offset;
    inv = 1.0f / *p.scale;
  }
  const long long w_id = (long long)blockIdx.x * kBandedWaves + wave;
  long long r = w_id * p.rows_per_wave;
  long long r_end = r + p.rows_per_wave;
  if (r_end > p.rows) r_end = p.rows;
  if (r >= r_end) return;
  typedef float v2f_ __attribute__((ext_vector_type(2)));
  auto fetch = [&](long long row, v2f_ (&v)[8], float& ny) {
    const v2f_* src = reinterpret_cast<const v2f_*>(p.X + row * 513);
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = src[lane + 64 * m];
    ny = reinterpret_cast<const float*>(src + 512)[0];      // broadcast: only the real part of the (real) Nyquist bin
  };
  auto walk = [&](long long row, const v2f_ (&v)[8], float ny) {
#pragma unroll
    for (int m = 0; m < 8; ++m) absrow[lane + 64 * m] = __builtin_amdgcn_sqrtf(fmaf(v[m].x, v[m].x, v[m].y * v[m].y));
    absrow[512 + lane] = (lane == 0) ? fabsf(ny) : 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float s0, s1;
    band_walk_fixed<FQ0, FQ1>(absrow, start0, start1, wlds, lane, s0, s1);
    const float y0 = (band_contrast_fast(s0, 1, 0.f) - off) * inv;
    const float y1 = (band_contrast_fast(s1, 1, 0.f) - off) * inv;
    float* orow = p.out + row * p.N;
    if (f0 >= 0) orow[f0] = y0;
    if (f1 >= 0) orow[f1] = y1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // two rows requested ahead of the one being walked (three register sets with rotating roles: 8 KB per wave in
  // flight; a request past the run's end re-reads its last row)
  v2f_ ra[8], rb[8], rc[8];
  float na, nb, nc;
  const long long last = r_end - 1;
  fetch(r, ra, na);
  fetch(r + 1 < r_end ? r + 1 : last, rb, nb);
  while (r < r_end) {
    fetch(r + 2 < r_end ? r + 2 : last, rc, nc);
    walk(r, ra, na);
    if (++r >= r_end) break;
    fetch(r + 2 < r_end ? r + 2 : last, ra, na);
    walk(r, rb, nb);
    if (++r >= r_end) break;
    fetch(r + 2 < r_end ? r + 2 : last, rb, nb);
    walk(r, rc, nc);
    ++r;
  }
}

static int launch_fixed_proj(const BandedParams& b, hipStream_t s) {
  FixedProjParams p = {};
  p.X = reinterpret_cast<const float2*>(b.A); p.out = b.out; p.offset = b.offset; p.scale = b.scale;
  p.weights = b.bank.weights; p.lane_start = b.bank.lane_start; p.lane_filter = b.bank.lane_filter;
  p.rows = b.rows; p.N = b.bank.n_filters;
  static thread_local long long slots = 0;
  if (!slots) {
    int dev = 0, cus = 256, nb = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mel_fixed_kernel<8, 2>, 64 * kBandedWaves, 0) != hipSuccess || nb <= 0) nb = 1;
    slots = (long long)cus * nb * kBandedWaves;
  }
  long long rpw = (p.rows + 4 * slots - 1) / (4 * slots);
  if (rpw < 8) rpw = 8;
  p.rows_per_wave = rpw;
  const long long waves = (p.rows + rpw - 1) / rpw;
  hipLaunchKernelGGL((mel_fixed_kernel<8, 2>), dim3((unsigned)((waves + kBandedWaves - 1) / kBandedWaves)), dim3(64 * kBandedWaves), 0, s, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

template <bool CPLX>
static int launch_banded(const BandedParams& p0, size_t dyn_lds, hipStream_t s) {
  BandedParams p = p0;
  const int nseg = (p.K + 63) / 64;
  void (*kernel)(BandedParams) = nullptr;
  int kseg = nseg;
  switch (nseg) {
    case 1: kernel = mel_banded_kernel<CPLX, 1, true>; break;
    case 2: kernel = mel_banded_kernel<CPLX, 2, true>; break;
    case 3: kernel = mel_banded_kernel<CPLX, 3, true>; break;
    case 4: kernel = mel_banded_kernel<CPLX, 4, true>; break;
    case 5: kernel = mel_banded_kernel<CPLX, 5, true>; break;
    case 6: kernel = mel_banded_kernel<CPLX, 6, true>; break;
    case 7: kernel = mel_banded_kernel<CPLX, 7, true>; break;
    case 8: kernel = mel_banded_kernel<CPLX, 8, true>; break;
    case 9: kernel = mel_banded_kernel<CPLX, 9, true>; break;
    case 10: kernel = mel_banded_kernel<CPLX, 10, true>; break;
    case 17: kernel = mel_banded_kernel<CPLX, 17, true>; break;       // n_fft 2048
    case 33: kernel = mel_banded_kernel<CPLX, 33, true>; break;       // n_fft 4096
    default:                                                          // in between: the next larger variant
      if (nseg < 17) { kernel = mel_banded_kernel<CPLX, 17, false>; kseg = 17; }
      else { kernel = mel_banded_kernel<CPLX, 33, false>; kseg = 33; }
      break;
  }
  // channel-major forward output of a one- or two-pass bank (MelSpectrogram / MFCC): the register-window variants
  if constexpr (CPLX) if (p.T > 0 && !p.inverse && !p.phase_out && !p.phase_in && p.bank.n_passes <= 2) {
    const bool two = p.bank.n_passes == 2;
    switch (nseg) {
#define CMV(N_) case N_: kernel = two ? mel_banded_kernel<CPLX, N_, true, 2> : mel_banded_kernel<CPLX, N_, true, 1>; break;
      CMV(3) CMV(5) CMV(9) CMV(17) CMV(33)
#undef CMV
      default: break;
    }
  }
  // LDS: table + lane tables + one row per wave.  The row holds the kernel's segments and whatever a walk that starts
  // on the row's last bins can run into.
  int max_walk = 0;
  for (int q = 0; q < p.bank.n_passes; ++q) max_walk = p.bank.pass_len[q] > max_walk ? p.bank.pass_len[q] : max_walk;
  // (idle lanes are parked on the first 16 quads of the row and walk like the others)
  const int reach = (p.K > 64 ? p.K : 64) + max_walk > 64 * kseg ? (p.K > 64 ? p.K : 64) + max_walk : 64 * kseg;
  p.row_floats = (reach + 63) / 64 * 64;
  int waves_per_block = kBandedWaves;
  while (waves_per_block > 1 && dyn_lds + (size_t)waves_per_block * p.row_floats * sizeof(float) > kLdsBudget) waves_per_block >>= 1;
  dyn_lds += (size_t)waves_per_block * p.row_floats * sizeof(float);
  if (dyn_lds > kLdsBudget) return AT_EUNSUPPORTED;
  // wave slots of this variant (occupancy x CUs), looked up once per (kernel, table size)
  struct Entry { const void* k; size_t lds; int dev; long long slots; };
  static thread_local Entry cache[16];
  static thread_local int n_cached = 0;
  long long slots = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return AT_ELAUNCH;
  for (int i = 0; i < n_cached; ++i)
    if (cache[i].k == (const void*)kernel && cache[i].lds == dyn_lds && cache[i].dev == dev) slots = cache[i].slots;
  if (!slots) {
    int cus = 256, nb = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    if (dyn_lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds) != hipSuccess) {
      (void)hipGetLastError();
      return AT_ELAUNCH;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 64 * waves_per_block, dyn_lds) != hipSuccess || nb <= 0) nb = 1;
    slots = (long long)cus * nb * waves_per_block;
    if (n_cached < 16) cache[n_cached++] = {(const void*)kernel, dyn_lds, dev, slots};
  }
  // a few whole rounds of resident waves: the table staging of a block is amortised and the tail stays short
  long long rpw = (p.rows + 4 * slots - 1) / (4 * slots);
  if (rpw < 8) rpw = 8;
  p.rows_per_wave = rpw;
  const long long waves = (p.rows + rpw - 1) / rpw;
  const long long blocks = (waves + waves_per_block - 1) / waves_per_block;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * waves_per_block), dyn_lds, s, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_mel_project_banded(const void* A, int a_kind, int64_t rows, int64_t lda, int K, const int32_t* lane_filter,
                          const int32_t* lane_start, const float* band_weights, int n_filters, int n_passes,
                          const int32_t* pass_len_host, int contrast, int inverse, const float* offset, const float* scale,
                          float eps, float* out, int64_t ld_out, int64_t T_transposed, float* phase_out, int64_t ld_phase,
                          const float* phase_offset, const float* phase_scale, const float* phase_in, void* stream) {
  if (rows < 0 || K <= 0 || n_filters <= 0) return AT_EINVAL;
  if (rows == 0) return AT_OK;
  if (!A || !out || !lane_filter || !lane_start || !band_weights || !pass_len_host) return AT_EINVAL;
  if (a_kind < 0 || a_kind > 3 || contrast < 0 || contrast > 3) return AT_EINVAL;
  if (n_passes <= 0 || n_passes > kMaxBandPasses || n_filters > 64 * n_passes) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (inverse && a_kind != A_REAL) return AT_EINVAL;
  if (phase_out && a_kind >= A_REAL) return AT_EINVAL;
  if (phase_in && (!inverse || phase_out || T_transposed)) return AT_EINVAL;
  if ((phase_offset == nullptr) != (phase_scale == nullptr)) return AT_EINVAL;
  if (K > kMaxRowK || (((uintptr_t)band_weights) & 15)) return AT_EUNSUPPORTED;
  BandedParams p = {};
  p.A = A; p.out = out; p.offset = offset; p.scale = scale;
  p.rows = rows; p.lda = lda; p.ld_out = ld_out; p.T = T_transposed;
  p.bank.lane_filter = lane_filter; p.bank.lane_start = lane_start; p.bank.weights = band_weights;
  p.bank.n_filters = n_filters; p.bank.n_passes = n_passes;
  p.K = K; p.a_kind = a_kind; p.contrast = contrast; p.inverse = inverse; p.eps = eps;
  p.phase_out = phase_out; p.ld_phase = ld_phase; p.ph_offset = phase_offset; p.ph_scale = phase_scale;
  p.phase_in = phase_in;
  size_t table_floats = 0;
  for (int q = 0; q < n_passes; ++q) {
    if (pass_len_host[q] < 0 || pass_len_host[q] > kMaxWalk || (pass_len_host[q] & 3)) return AT_EINVAL;
    p.bank.pass_len[q] = pass_len_host[q];
    table_floats += (size_t)64 * pass_len_host[q];
  }
  p.table_floats = (int)table_floats;
  const size_t dyn_lds = table_floats * sizeof(float) + (size_t)2 * 64 * n_passes * sizeof(int);
  hipStream_t s = (hipStream_t)stream;
  // the headline projection in its fixed form: |X| of 513-bin rows, log1p, two passes of 8 and 2 quads, row-major
  if (a_kind == A_COMPLEX_ABS && !inverse && contrast == C_LOG1P && K == 513 && lda == 513 && ld_out == n_filters &&
      T_transposed == 0 && !phase_out && !phase_in && n_passes == 2 && pass_len_host[0] == 32 && pass_len_host[1] == 8 &&
      (((uintptr_t)A) & 7) == 0 && variant(kVarEpilogue) == 0)
    return launch_fixed_proj(p, s);
  return a_kind >= A_REAL ? launch_banded<false>(p, dyn_lds, s) : launch_banded<true>(p, dyn_lds, s);
}

}  // extern "C"

// ---- small dense real projection: out = normalise(x @ W), K <= 128, N <= 64 (the DCT-II behind MFCC(n_mfcc)) ----
// One wavefront per row: lane n keeps column n of W in registers for the whole launch (K VGPRs), the row is
// broadcast from LDS four values per ds_read_b128, K multiply-adds per lane.  The MFMA projection of mel.hip
// wastes most of a 128-column block on such a narrow matrix (0.56 ms for 706 560 rows of 128 -> 40; this: HBM).
namespace at_hip {

struct SmallProjParams {
  const float* x;       // rows x K
  const float* W;       // K x N row-major
  float* out;
  const float* offset;
  const float* scale;
  long long rows, T;    // T > 0: channel-major store out[(r / T) * N * T + n * T + r % T]
  long long rows_per_wave;
  int K, N;
};

template <int KQ>   // KQ = ceil(K / 4) quads of the row, <= 32
__global__ __launch_bounds__(256) void small_proj_kernel(SmallProjParams p) {
  __shared__ float4 rowbuf[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f w[2 * KQ];       // this lane's column of W, two rows per register pair (packed multiply-adds below)
#pragma unroll
  for (int k = 0; k < 4 * KQ; ++k) w[k >> 1][k & 1] = (k < p.K && lane < p.N) ? p.W[(long long)k * p.N + lane] : 0.0f;
  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }
  long long r = ((long long)blockIdx.x * 4 + wave) * p.rows_per_wave;
  long long r_end = r + p.rows_per_wave;
  if (r_end > p.rows) r_end = p.rows;
  float* buf = reinterpret_cast<float*>(rowbuf[wave]);
  // channel-major output: every lane owns one row of the (.., N, T) tensor; eight frames are kept in registers
  // and stored as 32 contiguous bytes (a 4-byte store per frame leaves partly written lines to be re-fetched)
  float cm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cm[k] = 0.f;
  long long cb = (p.T > 0) ? r / p.T : 0, ct = (p.T > 0) ? r - cb * p.T : 0;
  long long run_t0 = ct;                          // first frame of this wave's rows inside the current clip
  const int k0 = 2 * lane;              // a row is K <= 128 floats: two per lane
  // Rows r .. r+3 in flight in a ring of four register pairs with compile-time roles (copying a register whose load is
  // still out waits for it, and a conditional load makes the compiler drain vmcnt on the spot: either way the
  // "prefetch" degenerates to one row).  Loads are unconditional: rows past the end re-read the last row, floats past
  // K re-read element K-1 and meet a zero weight.
  const int ka = k0 < p.K ? k0 : p.K - 1, kb = k0 + 1 < p.K ? k0 + 1 : p.K - 1;
  const long long last_row = r_end - 1;
  auto fetch = [&](long long row, float& a, float& b) {
    const float* src = p.x + (row < last_row ? row : last_row) * p.K;
    a = src[ka];
    b = src[kb];
  };
  float ra[4], rb[4];
  if (r >= r_end) return;
#pragma unroll
  for (int d = 0; d < 4; ++d) fetch(r + d, ra[d], rb[d]);
  auto row_step = [&](float& slot_a, float& slot_b) {
    // stage the current row, then refill its slot with row r+4
    buf[k0] = slot_a;
    buf[k0 + 1] = slot_b;
    fetch(r + 4, slot_a, slot_b);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v2f a01 = {0.f, 0.f}, a23 = {0.f, 0.f};          // four independent chains (k mod 4), two per packed fma
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const float4 v = rowbuf[wave][q];            // same address on every lane: LDS broadcast
      a01 = __builtin_elementwise_fma((v2f){v.x, v.y}, w[2 * q], a01);
      a23 = __builtin_elementwise_fma((v2f){v.z, v.w}, w[2 * q + 1], a23);
    }
    float acc = (a01.x + a01.y) + (a23.x + a23.y);
    if (p.offset) acc = (acc - off) / sc;
    if (p.T > 0) {
      const long long b = cb, t = ct;               // (clip, frame) of row r, tracked incrementally
      if (++ct == p.T) {
        ct = 0;
        ++cb;
      }
      if (t == 0) run_t0 = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) cm[k] = cm[k + 1];
      cm[7] = acc;
      if (((t & 7) == 7 || t == p.T - 1 || r == r_end - 1) && lane < p.N) {
        long long first = t & ~7LL;                 // frames [first, t] are new since the last flush
        if (first < run_t0) first = run_t0;
        float* dst = p.out + (b * p.N + lane) * p.T + (t - 7);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (t - 7 + k >= first) dst[k] = cm[k];
      }
    } else if (lane < p.N) {
      p.out[r * p.N + lane] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    ++r;
  };
  while (r < r_end) {
    row_step(ra[0], rb[0]);
    if (r >= r_end) break;
    row_step(ra[1], rb[1]);
    if (r >= r_end) break;
    row_step(ra[2], rb[2]);
    if (r >= r_end) break;
    row_step(ra[3], rb[3]);
  }
}

// The same contraction on the matrix cores, for K = 128 / 80 / 64 and 16-byte aligned rows (MFCC: 128 log-mel values ->
// 40 coefficients, 706 560 rows).  The kernel above is bound by its LDS broadcasts (32 ds_read_b128 per row per wave:
// 0.343 ms against 0.09 ms of HBM traffic); here the rows never pass through LDS.  out^T = W^T x^T in tiles of
// 16 channels x 16 rows with v_mfma_f32_16x16x4_f32 (fp32 products, fp32 accumulation -- no reduced precision): the A
// operand is W^T (lane (m, g) supplies W[k][16 mt + m] for its K/4 values of k), the B operand is x^T -- lane (m, g)
// supplies x[row m][k] for the same k, which it loads itself from global memory.  The contraction order is free, so step
// (j, c) takes k = 16 j + 4 g + c: a lane's operands are the float4s at k = 16 j + 4 g, four lanes cover 64 contiguous
// bytes of a row and every byte of the tile is requested exactly once.  MT = ceil(N / 16) channel tiles (40 -> 3: 17 %
// of the MFMA work is padding), KS = K / 16.  0.133 ms (A/Bs in profiles/r03h_small_projection.md); the floor of the
// MFMA pipe is 0.055 ms at 2.4 GHz, of the memory system 0.09 ms.
typedef float v4f_t __attribute__((ext_vector_type(4)));

// WLDS: the A operands are read from LDS (one ds_read_b128 per (mt, j): 24 KB per tile and wave, a sixth of what the
// LDS delivers in the tile's MFMA time) instead of being held in 32 MT KS / 8 registers: four waves per SIMD instead of two.
template <int KS, int MT, bool WLDS, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WLDS ? 4 : 2))) void small_proj_mfma_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              SmallProjParams p) {
  __shared__ v4f_t wlds[WLDS ? MT * KS * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, g = lane >> 4;
  float w[WLDS ? 1 : MT][WLDS ? 1 : KS][4];
  if (WLDS) {
    // wave w fills the (mt, j) slots w, w+4, ..: every lane writes its own operands
    for (int q = wave; q < MT * KS; q += 4) {
      const int mt = q / KS, j = q - mt * KS;
      const int ch = 16 * mt + m;
      v4f_t v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = ch < p.N ? p.W[(long long)(16 * j + 4 * g + c) * p.N + ch] : 0.0f;
      wlds[q * 64 + lane] = v;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int mt = 0; mt < (WLDS ? 1 : MT); ++mt)
#pragma unroll
      for (int j = 0; j < (WLDS ? 1 : KS); ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ch = 16 * mt + m, k = 16 * j + 4 * g + c;
          w[mt][j][c] = ch < p.N ? p.W[(long long)k * p.N + ch] : 0.0f;
        }
  }
  float off = 0.f, sc = 1.f;
  if (p.offset) {
    off = *p.offset;
    sc = *p.scale;
  }
  long long r = ((long long)blockIdx.x * 4 + wave) * p.rows_per_wave;      // rows_per_wave is a multiple of 32
  long long r_end = r + p.rows_per_wave;
  if (r_end > p.rows) r_end = p.rows;
  if (r >= r_end) return;
  const long long last_row = r_end - 1;
  const int K = 16 * KS;
  // loads are unconditional: tile rows past the end re-read the last row and are not stored
  auto fetch = [&](long long r0, float4 (&v)[KS]) {
    const long long row = r0 + m < last_row ? r0 + m : last_row;
    const float4* src = reinterpret_cast<const float4*>(x + row * K) + g;
#pragma unroll
    for (int j = 0; j < KS; ++j) v[j] = src[4 * j];
  };
  auto contract = [&](const float4 (&v)[KS], v4f_t (&acc)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const float xs[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      if (WLDS) {
        // volatile: the table does not change, and a plain read is hoisted out of the row loop into MT KS x 4 registers
        typedef const volatile __attribute__((address_space(3))) v4f_t* lds_ptr;
        v4f_t wq[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wq[mt] = *(lds_ptr)(&wlds[(mt * KS + j) * 64 + lane]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[mt][c], xs[c], acc[mt], 0, 0, 0);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[WLDS ? 0 : mt][WLDS ? 0 : j][c], xs[c], acc[mt], 0, 0, 0);
      }
    }
  };
  // Two tiles are stored together.  A tile's result has 16 rows on the lanes of each 16-lane group and the group's four
  // channels in the registers: stored as it is, a channel receives 64 contiguous bytes.  v_permlane16_swap exchanges the
  // odd groups of the first tile's register with the even groups of the second tile's: afterwards lanes 0..31 (32..63)
  // hold ONE channel of 32 consecutive rows -- 128 contiguous bytes per channel and store.
  const int rr = lane & 31, hi = lane >> 5;
  long long cb = 0, ct = 0;                 // (clip, frame) of the row this lane stores
  if (p.T > 0) {
    cb = (r + rr) / p.T;
    ct = (r + rr) - cb * p.T;
  }
  auto store_pair = [&](const v4f_t (&a0)[MT], const v4f_t (&a1)[MT]) {
    const bool row_ok = r + rr <= last_row;
    float* base = p.T > 0 ? out + cb * p.N * p.T + ct : out + (r + rr) * p.N;
    const long long ch_stride = p.T > 0 ? p.T : 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0[mt][i]), __float_as_uint(a1[mt][i]), false, false);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = 16 * mt + 8 * hi + 4 * h + i;
          float a = __uint_as_float(sw[h]);
          if (p.offset) a = (a - off) / sc;
          if (row_ok && ch < p.N) {
            float* dst = base + ch * ch_stride;
            if (NT) __builtin_nontemporal_store(a, dst);
            else *dst = a;
          }
        }
      }
    r += 32;
    if (p.T > 0) {
      ct += 32;
      while (ct >= p.T) {
        ct -= p.T;
        ++cb;
      }
    }
  };
  float4 va[KS], vb[KS];
  v4f_t acc0[MT], acc1[MT];
  fetch(r, va);
  while (true) {
    fetch(r + 16, vb);
    contract(va, acc0);
    fetch(r + 32, va);
    contract(vb, acc1);
    store_pair(acc0, acc1);
    if (r >= r_end) break;
  }
}

template <int KS, bool WLDS, bool NT>
static bool launch_small_mfma(const SmallProjParams& p, int mt, dim3 grid, hipStream_t s) {
  const dim3 block(256);
  switch (mt) {
    case 1: hipLaunchKernelGGL((small_proj_mfma_kernel<KS, 1, WLDS, NT>), grid, block, 0, s, p.x, p.out, p); return true;
    case 2: hipLaunchKernelGGL((small_proj_mfma_kernel<KS, 2, WLDS, NT>), grid, block, 0, s, p.x, p.out, p); return true;
    case 3: hipLaunchKernelGGL((small_proj_mfma_kernel<KS, 3, WLDS, NT>), grid, block, 0, s, p.x, p.out, p); return true;
    case 4: hipLaunchKernelGGL((small_proj_mfma_kernel<KS, 4, WLDS, NT>), grid, block, 0, s, p.x, p.out, p); return true;
  }
  return false;
}

}  // namespace at_hip

extern "C" int at_project_small(const float* x, int64_t rows, int K, const float* W, int N, const float* offset,
                                const float* scale, float* out, int64_t T_transposed, void* stream) {
  using namespace at_hip;
  if (rows < 0 || K <= 0 || N <= 0) return AT_EINVAL;
  if (rows == 0) return AT_OK;
  if (!x || !W || !out) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (K > 128 || N > 64) return AT_EUNSUPPORTED;
  SmallProjParams p = {x, W, out, offset, scale, rows, T_transposed, 0, K, N};
  hipStream_t s = (hipStream_t)stream;
  if ((K == 128 || K == 80 || K == 64) && (((uintptr_t)x) & 15) == 0 && variant(kVarSmallProjection) == 0) {
    // matrix-core form: every wave a whole number of 32-row tile pairs
    const char* form = dev_env("ACIDS_PROJECT_SMALL_FORM");     // dev builds: "regs", "regs_nt", "lds", "lds_nt"
    const int f = !form ? 2 : !strcmp(form, "regs") ? 0 : !strcmp(form, "regs_nt") ? 1 : !strcmp(form, "lds") ? 2 : 3;
    const long long waves_target = (long long)num_cus() * (f >= 2 ? 16 : 8);
    long long rpw = (rows + waves_target - 1) / waves_target;
    rpw = (rpw + 31) / 32 * 32;
    if (rpw < 64) rpw = 64;
    p.rows_per_wave = rpw;
    const long long waves = (rows + rpw - 1) / rpw;
    const dim3 grid((unsigned)((waves + 3) / 4));
    const int mt = (N + 15) / 16;
    bool ok;
    if (K == 128 && f == 0) ok = launch_small_mfma<8, false, false>(p, mt, grid, s);
    else if (K == 128 && f == 1) ok = launch_small_mfma<8, false, true>(p, mt, grid, s);
    else if (K == 128 && f == 2) ok = launch_small_mfma<8, true, false>(p, mt, grid, s);
    else if (K == 128) ok = launch_small_mfma<8, true, true>(p, mt, grid, s);
    else if (K == 80) ok = launch_small_mfma<5, true, true>(p, mt, grid, s);
    else ok = launch_small_mfma<4, true, true>(p, mt, grid, s);
    if (!ok) return AT_EINVAL;
    return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
  }
  const long long waves_target = 256LL * 32;
  long long rpw = (rows + waves_target - 1) / waves_target;
  if (rpw < 4) rpw = 4;
  p.rows_per_wave = rpw;
  const long long waves = (rows + rpw - 1) / rpw;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  const int kq = (K + 3) / 4;
  if (kq <= 8) hipLaunchKernelGGL(small_proj_kernel<8>, grid, block, 0, s, p);
  else if (kq <= 16) hipLaunchKernelGGL(small_proj_kernel<16>, grid, block, 0, s, p);
  else hipLaunchKernelGGL(small_proj_kernel<32>, grid, block, 0, s, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}
