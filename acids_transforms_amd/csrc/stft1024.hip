// stft1024.hip -- fused framing + window + rFFT-1024 (forward) and
// irFFT-1024 + window + overlap-add (inverse) for gfx950.
//
// Replaces, for n_fft = 1024:
//   torch.stft(...).transpose(-2,-1)        reference transforms/stft.py:98-104, dgt.py:64-70   (K1)
//   x_fft.angle() phase buffer              stft.py:103, dgt.py:69                               (K2, optional)
//   torch.istft(...)                        stft.py:120-128, dgt.py:86-93                        (K3)
//   x * exp(1j*phase) before istft          stft.py:157-161, dgt.py:152-154                      (K15)
//   torch.fft.rfft(x*window) / irfft(x)*w   stft.py:249-266, dgt.py:285-302 (pre-framed)         (K4/K5)
//
// Execution model: one 64-lane wavefront owns one frame at a time and keeps
// window, twiddles (and, for the inverse, the overlap-add accumulators) in
// registers across a run of frames.  No workgroup barrier anywhere; a
// 256-thread block is just four independent waves sharing an LDS allocation.
#include <hip/hip_runtime.h>
#include "fastmath.h"
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <atomic>

#include "band_bank.h"
#include "fft512.h"
#include "run_plan.h"
#include "variants.h"

#ifndef AT_ISTFT_NT
#define AT_ISTFT_NT 1
#endif
#ifndef AT_ISTFT_NTLOAD
#define AT_ISTFT_NTLOAD 1      // spectra are read once: -1 ... -2 % (alternating A/B, tools/ab3.sh)
#endif

namespace at_hip {

constexpr int N = 1024;
constexpr int F = 513;
constexpr int WAVES_PER_BLOCK = 4;

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
struct FwdParams {
  const float* x;       // audio, clip b at x + b*clip_stride
  const float* window;  // N analysis window samples
  const float2* tw;     // twiddle table (fft512.h)
  float2* out;          // (B*T, 513) complex64
  float* phase;         // (B*T, 513) or nullptr
  long long B, L, clip_stride, T;
  long long total_frames;     // B*T
  long long frames_per_block; // multiple of WAVES_PER_BLOCK
  int hop;
  int center;  // 1: torch.stft center=True/reflect; 0: frame t starts at t*hop
};

__device__ __forceinline__ long long reflect_index(long long i, long long L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// loads z[lane + 64 m] = (x[2n], x[2n+1]) of frame (b, t) into v
__device__ __forceinline__ void load_frame(const FwdParams& p, long long f, int lane, float2 (&v)[8]) {
  const long long b = f / p.T;
  const long long t = f - b * p.T;
  const float* clip = p.x + b * p.clip_stride;
  const long long start = t * (long long)p.hop - (p.center ? N / 2 : 0);
  const bool interior = (start >= 0) && (start + N <= p.L);
  const bool aligned = (((uintptr_t)(clip + start)) & 7) == 0;
  if (interior && aligned) {
    const float2* src = reinterpret_cast<const float2*>(clip + start);
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = src[lane + 64 * m];
  } else if (interior) {
    const float* src = clip + start;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      v[m].x = src[2 * (lane + 64 * m)];
      v[m].y = src[2 * (lane + 64 * m) + 1];
    }
  } else {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      long long i0 = start + 2 * (lane + 64 * m);
      if (p.center) {
        v[m].x = clip[reflect_index(i0, p.L)];
        v[m].y = clip[reflect_index(i0 + 1, p.L)];
      } else {  // zero padding past the end (utils/misc.py:156 pad())
        v[m].x = (i0 < p.L) ? clip[i0] : 0.0f;
        v[m].y = (i0 + 1 < p.L) ? clip[i0 + 1] : 0.0f;
      }
    }
  }
}

template <bool WRITE_PHASE>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void stft1024_fwd_kernel(FwdParams p) {
  __shared__ float2 lds_all[WAVES_PER_BLOCK * kFftLdsFloat2PerWave];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;

  Twiddles tw;
  load_twiddles<false>(tw, p.tw, lane);
  float2 win[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) win[m] = reinterpret_cast<const float2*>(p.window)[lane + 64 * m];

  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;

  long long f = f_begin + wave;
  float2 nxt[8];
  if (f < f_end) load_frame(p, f, lane, nxt);
  for (; f < f_end; f += WAVES_PER_BLOCK) {
    float2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = make_float2(nxt[m].x * win[m].x, nxt[m].y * win[m].y);
    // software prefetch of this wave's next frame
    if (f + WAVES_PER_BLOCK < f_end) load_frame(p, f + WAVES_PER_BLOCK, lane, nxt);

    fft512<false>(v, tw, lds, lane);
    float2 nyq;
    rfft_merge(v, tw, lane, nyq);

    float2* row = p.out + f * F;
#pragma unroll
    for (int m = 0; m < 8; ++m) row[lane + 64 * m] = v[m];
    if (lane == 0) row[512] = nyq;
    if (WRITE_PHASE) {
      float* prow = p.phase + f * F;
#pragma unroll
      for (int m = 0; m < 8; ++m) prow[lane + 64 * m] = fast_atan2f(v[m].y, v[m].x);
      if (lane == 0) prow[512] = fast_atan2f(nyq.y, nyq.x);
    }
  }
}

// ---------------------------------------------------------------------------
// forward, hop = 256 = N/4, center=True: sliding-window variant.
// A wave walks a run of consecutive frames of one clip.  Frame t+1 shares 768 of
// its 1024 samples with frame t, and in the lane layout z[l + 64 m] that overlap
// is exactly "registers m+2 of the same lane": the raw samples stay in registers,
// shifted by two slots per frame, and only the 256 new samples (two 8-byte loads
// per lane) are fetched per frame -- 1 KB of loads per frame instead of 4 KB.
// ---------------------------------------------------------------------------
struct FwdRunParams {
  const float* x;
  const float* window;
  const float2* tw;
  float2* out;     // may be null when only the features are wanted (MEL == 2)
  float* phase;
  long long B, L, clip_stride, T;
  long long runs_per_clip, frames_per_run;
  // fused magnitude / mel epilogue (MEL != 0)
  BandBank bank;
  float* feat;            // (B*T, N) or, with feat_channel_major, (B, N, T)
  const float* offset;    // device scalars or null
  const float* scale;
  float eps;
  int contrast;           // 0 none, 1 log1p, 2 log, 3 log10
  int power2;             // |X|^2 instead of |X|
  int feat_channel_major;
  // POLAR: features and normalise(angle) rows written side by side into a stacked (B*T, 2, F) tensor
  long long feat_ld, phase_ld;   // row strides (0: n_filters / F)
  const float* ph_offset;        // Normalize affine of the phase half (device scalars or null)
  const float* ph_scale;
  // PW (persistent workgroups): tiles of FWD_WAVES consecutive runs are handed out in address order
  unsigned* tile_ctr;            // {next tile, workgroups done}: zero before the launch, reset by the last workgroup
  unsigned n_tiles;
};

// element n = lane + 64 m of the frame starting at padded position p0 (original index p0 - 512 + 2n)
__device__ __forceinline__ float2 load_pair(const float* clip, long long L, long long i0, bool interior, bool aligned,
                                            int lane_off) {
  // i0: original index of the first sample of this 64-lane, 128-sample segment; lane_off = 2*lane
  if (interior) {
    if (aligned) return reinterpret_cast<const float2*>(clip + i0)[lane_off >> 1];
    return make_float2(clip[i0 + lane_off], clip[i0 + lane_off + 1]);
  }
  return make_float2(clip[reflect_index(i0 + lane_off, L)], clip[reflect_index(i0 + lane_off + 1, L)]);
}


// Contrast of the fused epilogue.  The arguments are >= eps = 1.19e-7 (never denormal), so the hardware log2
// (1 ulp) times ln 2 / log10 2 is within ~2 ulp of logf / log10f at a sixth of the instructions.
__device__ __forceinline__ float fwd_contrast(float v, int mode, float eps) { return band_contrast_fast(v, mode, eps); }

// MEL: 0 = spectrum only; 1 = spectrum + fused banded-filterbank features; 2 = features only.
// FWD_WAVES waves per workgroup share the LDS constant tables (twiddles are always staged there; TWLDS makes
// the FFT read them at the point of use instead of holding 44 VGPRs, which buys a fourth wave per SIMD).
// CMBUF (MEL != 0, channel-major features): number of passes whose outputs are kept for eight frames in
// registers and written as 32 contiguous bytes per filter; 0 = every frame scatters 4-byte stores (each
// lane its own row of the (B, N, T) tensor), which leaves partly written lines to be evicted and re-fetched.
// POLAR (with MEL == 2): besides the features, normalise(angle X) of every bin goes to p.phase with row stride
// p.phase_ld -- Compose(STFT + Polar) in one kernel, the complex spectrum never reaches HBM.
// HS: hop in 128-sample register slots (1, 2 = the reference's default hop 256, 4): the window slides HS slots per
// frame and HS new segments are fetched.
// SP (row-major features of a one- or two-pass bank -- the 128-mel bank of the headline step): the passes are
// unrolled and what a lane needs for them (its filter, where its walk starts, the walk's length) is read once per
// run instead of once per pass and frame.
// FQ0 / FQ1 (with SP == 2): the two passes' walk lengths in quads as compile-time constants, log1p contrast and
// |X| (not |X|^2) fixed -- the headline configuration (128 mel filters at 44.1 kHz: 8 and 2 quads).  The generic
// epilogue spends more instructions on run-time switches (contrast mode, power, layout, loop control: 112 scalar
// and 137 vector instructions per frame in the listing) than on the 20 multiply-adds of the walk itself; with
// everything fixed both passes are straight-line code, their LDS reads batched and their sums independent.
// AL: the spectrum leaves as ONE byte stream in 512-byte aligned blocks.  (B, T, 513) complex64 is contiguous and a
// wave writes consecutive frames, but a row is 4104 bytes: row f starts 8 f bytes past a 128-byte line, every one of its
// eight 512-byte stores straddles five lines and the Nyquist bin is a ninth, one-lane store.  With AL the output
// COLUMNS of the FFT are rotated over the lanes by rot = (f 513) mod 64 (free: the last exchange reads through the
// rotated index, fft512's out_lane), so that the lane number IS the position inside an aligned block of 64 bins: lanes
// >= rot hold block j of the frame in register j, lanes < rot hold block j + 1 in register j, block 8 (the tail of
// register 7, then the Nyquist bin on lane rot) is carried into the next frame's block 0.  Eight full, aligned 512-byte
// stores per frame (a ninth every 64 frames), two selects per store, no masked store in the steady state.
// NT: those stores non-temporal.  tools/ubench/stream_pattern2.hip prices the pattern: rows 4.7 TB/s, aligned blocks
// 4.95, aligned + nt 5.0-5.3 (profiles/r03a_*).
template <bool WRITE_PHASE, int MEL, int FWD_WAVES, bool TWLDS, int CMBUF = 0, bool POLAR = false, int HS = 2, int SP = 0,
          int FQ0 = 0, int FQ1 = 0, bool AL = false, bool NT = false, int HYB = 0, int FC = 1, bool FP2 = false, bool PW = false,
          fqp_t FQP = 0, int FNP = 0>
__global__ __launch_bounds__(64 * FWD_WAVES, (TWLDS && !CMBUF && !HYB) ? 4 : 3) void stft1024_h256_fwd_kernel(FwdRunParams p) {
  // HYB (with TWLDS): bit 0 -- the two pass twiddle tables in registers, only the merge's W1024 rows from LDS
  // (HybridTwiddles); bit 1 -- the analysis window in registers.  Both trade LDS reads (the busiest unit of these
  // kernels) for VGPRs, i.e. for the fourth wave per SIMD.
  constexpr int H = 128 * HS;
  constexpr int kNP = FQP ? FNP : 1;
  constexpr int kPkStride = (kNP + 3) / 4 * 4;       // dwords per lane in the packed descriptor table (16-byte rows)
  constexpr int kTabTw = TWLDS ? kTwiddleCount : 0;
  // FQP kernels: 1 KB per wave IN FRONT of its FFT slab -- the floats of the feature stream that are carried into the
  // next frame's first aligned block (the slab itself is rewritten by every transform)
  constexpr int kFeatCarry = FQP ? 128 : 0;               // float2
  constexpr int kWaveLds = kFftLdsFloat2PerWave + kFeatCarry;
  __shared__ float2 lds_all[FWD_WAVES * kWaveLds + kTabTw + 512];
  extern __shared__ float4 band_lds[];   // MEL != 0: the bank's weight table, sized by the launcher
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; said explicitly, or everything derived from it (clip, run bounds, the frame
  // counter of the main loop, the column rotation) lives in vector registers and the loop control runs on the VALU
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float2* lds = lds_all + wave * kWaveLds + kFeatCarry;
  float2* tab = lds_all + FWD_WAVES * kWaveLds;
  // (non-persistent launches) the eight segments of this wave's first frame are requested BEFORE the workgroup stages its
  // LDS tables: their latency (cold HBM) then runs under the table fill and the barrier instead of after them -- part of
  // what a run start costs (profiles/r04_launch_shape.md (c)), and what short runs in dispatch order have to pay per run
  float2 raw_first[8];
  auto load_first_frame = [&](const long long run, float2 (&raw)[8]) {
    const long long b = run / p.runs_per_clip;
    if (b >= p.B) return;
    const long long t0 = (run - b * p.runs_per_clip) * p.frames_per_run;
    if (t0 >= p.T) return;
    const float* clip = p.x + b * p.clip_stride;
    const bool clip_aligned = ((((uintptr_t)clip) & 7) == 0);
    const long long s0 = t0 * (128 * HS) - 512;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const long long i0 = s0 + 128 * m;
      raw[m] = load_pair(clip, p.L, i0, i0 >= 0 && i0 + 128 <= p.L, clip_aligned, 2 * lane);
    }
  };
  if constexpr (!PW) load_first_frame((long long)blockIdx.x * FWD_WAVES + wave, raw_first);
  // workgroup-shared constants: (twiddle table,) analysis window, band weights
  if (TWLDS)
    for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * FWD_WAVES) tab[i] = twiddle_for_lds<false>(p.tw, i);
  for (int i = threadIdx.x; i < 512; i += 64 * FWD_WAVES)
    tab[kTabTw + i] = reinterpret_cast<const float2*>(p.window)[i];
  float* wlds = reinterpret_cast<float*>(band_lds);
  int* lane_tab = nullptr;
  if (MEL != 0) {
    int table_floats = 0;
    for (int q = 0; q < p.bank.n_passes; ++q) table_floats += 64 * p.bank.pass_len[q];
    for (int i = threadIdx.x; i < table_floats; i += 64 * FWD_WAVES) wlds[i] = p.bank.weights[i];
    // per-lane walk descriptors behind the weights: re-read each frame (two ds_read_b32 per pass) rather
    // than held in two VGPRs per pass (up to 32 with a 16-pass bank)
    lane_tab = reinterpret_cast<int*>(wlds + table_floats);
    for (int i = threadIdx.x; i < 64 * p.bank.n_passes; i += 64 * FWD_WAVES) {
      lane_tab[i] = p.bank.lane_start[i];
      lane_tab[64 * p.bank.n_passes + i] = p.bank.lane_filter[i];
    }
    // FQP (fixed-length epilogue of a bank with many passes, band_bank.h): ONE word per lane and pass -- the byte offset
    // of the lane's walk in the LDS row (low 12 bits) and the byte offset of its filter in the feature row (next 12;
    // 0xfff: no filter) -- lane-major behind the lane tables, read back 16 bytes at a time (rows of kPkStride words: 12
    // for nine passes, which keeps the 16 lanes of a ds_read_b128 group on distinct banks).  In registers the nine words
    // (and what the optimiser hoists out of the frame loop from them) spilled.
    if constexpr (FQP != 0) {
      int* pk_tab = lane_tab + 2 * 64 * p.bank.n_passes;
      for (int i = threadIdx.x; i < 64 * kNP; i += 64 * FWD_WAVES) {
        const int q = i >> 6, l = i & 63;
        const int f = p.bank.lane_filter[i];
        pk_tab[l * kPkStride + q] = (p.bank.lane_start[i] * 4) | ((f >= 0 ? f * 4 : 0xfff) << 12);
      }
    }
  }
  // PW: the workgroup stays resident and takes TILES -- FWD_WAVES consecutive runs, one per wave -- in address order from
  // a device counter (one returning atomic per workgroup and tile, requested one tile ahead), with a workgroup barrier
  // per tile.  Short runs (8 frames) then cost neither a workgroup launch nor an LDS table refill, and the chip writes
  // ONE advancing window of ~4096 short runs instead of 4096 fronts a run length apart: the access-pattern ceiling of
  // the fused forward rises from 5.06 to 5.6-5.8 TB/s (tools/ubench/stream_pattern3.hip, shapes P / D / Pw;
  // profiles/r04_launch_shape.md).
  __shared__ unsigned s_tile[2];
  if (PW && threadIdx.x == 0) s_tile[0] = atomicAdd(p.tile_ctr, 1u);
#if AT_XCHG1_SWAP
  // with the first exchange in registers only the second one rewrites the slab: the floats behind bin 512 that it does not
  // reach must be finite for the fixed-length epilogues (fft512.h) -- cleared once
  if (MEL != 0) {
    reinterpret_cast<float*>(lds)[512 + lane] = 0.0f;
    reinterpret_cast<float*>(lds)[576 + lane] = 0.0f;
  }
#endif
  __syncthreads();

  int sp_f[SP > 0 ? SP : 1], sp_start[SP > 0 ? SP : 1], sp_quads[SP > 0 ? SP : 1];
  if constexpr (SP > 0) {
#pragma unroll
    for (int q = 0; q < SP; ++q) {
      sp_f[q] = lane_tab[(SP + q) * 64 + lane];
      sp_start[q] = lane_tab[q * 64 + lane];
      sp_quads[q] = p.bank.pass_len[q] >> 2;
    }
  }
  Twiddles tw_regs;
  if (!TWLDS) load_twiddles<false>(tw_regs, p.tw, lane);
  const LdsTwiddles<false> tw_lds = {tab, lane};
  HybridTwiddles tw_hyb;
  if constexpr (TWLDS && (HYB & 1)) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      tw_hyb.t1[k] = to_v(p.tw[k * 64 + lane]);
      tw_hyb.t2[k] = to_v(p.tw[(7 + k) * 64 + lane]);
    }
    tw_hyb.tab = tab;
    tw_hyb.col = lane;
  }
  const float2* win = tab + kTabTw;   // win[lane + 64 m]
  v2f win_regs[8];
  if constexpr ((HYB & 2) != 0) {
#pragma unroll
    for (int m = 0; m < 8; ++m) win_regs[m] = to_v(reinterpret_cast<const float2*>(p.window)[lane + 64 * m]);
  }

  const long long L = p.L;
  const int lane2 = 2 * lane;

  auto do_run = [&](const long long run, const bool preloaded) {
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long t0 = r * p.frames_per_run;
  long long t1 = t0 + p.frames_per_run;
  if (t1 > p.T) t1 = p.T;
  if (t0 >= t1) return;
  const float* clip = p.x + b * p.clip_stride;
  const bool clip_aligned = ((((uintptr_t)clip) & 7) == 0);  // frame starts are multiples of 256 samples

  // first frame of the run: all eight segments (already requested above unless the workgroup is persistent)
  float2 raw[8];
  if (preloaded) {
#pragma unroll
    for (int m = 0; m < 8; ++m) raw[m] = raw_first[m];
  } else {
    load_first_frame(run, raw);
  }
  // Make the prologue loads architecturally complete here: otherwise the loop-top wait is the merge of
  // "just loaded" (from this prologue) and "loaded nine stores ago" (from the back edge) and collapses to
  // vmcnt(0), which would drain every frame's stores before the next frame starts.
#pragma unroll
  for (int m = 0; m < 8; ++m) asm volatile("" : "+v"(raw[m].x), "+v"(raw[m].y));
  float2* row = p.out + (b * p.T + t0) * F;
  const long long feat_ld = (POLAR && p.feat_ld) ? p.feat_ld : (long long)p.bank.n_filters;
  const long long phase_ld = (POLAR && p.phase_ld) ? p.phase_ld : F;
  float* prow = (WRITE_PHASE || POLAR) ? p.phase + (b * p.T + t0) * phase_ld : nullptr;
  float* frow = (MEL != 0) ? p.feat + (b * p.T + t0) * feat_ld : nullptr;
  // FQP: the features leave as ONE stream of 1-KB aligned blocks too ((B, T, 513) floats are contiguous; a row is 2052
  // bytes): fc floats of the current block are already in hand (carried in LDS), fdst is that block, fhead the number of
  // leading floats of the run's first block that belong to the run before it
  int fc = 0, fhead = 0;
  float* fdst = nullptr;
  if constexpr (FQP != 0) {
    const long long fe0 = (b * p.T + t0) * (long long)F;
    fc = (int)(fe0 & 255);
    fhead = fc;
    fdst = p.feat + (fe0 - fc);
  }
  float ph_off = 0.f, ph_sc = 1.f;
  if (POLAR && p.ph_offset) {
    ph_off = *p.ph_offset;
    ph_sc = *p.ph_scale;
  }
  int rot = 0;                 // AL: column rotation of the current frame
  float2* sp = nullptr;        //     this lane's slot in the current frame's block 0
  v2f carry = {0.f, 0.f};      //     block 8 of the previous frame (lanes < rot: bins 448 + .., lane rot - 1: Nyquist)
  bool head = true;            //     the run's first block belongs partly to the run before
  if constexpr (AL) {
    static_assert(MEL != 2 && !WRITE_PHASE && !POLAR && TWLDS, "aligned stream stores: spectrum out, LDS twiddles");
    const long long e0 = (b * p.T + t0) * F;
    rot = (int)(e0 & 63);
    sp = p.out + (e0 - rot) + lane;
  }
  auto put = [&](float2* dst, v2f val) {
    if (NT) __builtin_nontemporal_store(val, reinterpret_cast<v2f*>(dst));
    else *reinterpret_cast<v2f*>(dst) = val;
  };
  long long t_cur = t0;
  float cm[CMBUF ? CMBUF : 1][8];   // features of frames t_cur-7 .. t_cur (sliding), one row per pass
  int cm_held[CMBUF ? CMBUF : 1];   // frames of the window that are not stored yet (per lane: flush points differ)
#pragma unroll
  for (int q = 0; q < (CMBUF ? CMBUF : 1); ++q) {
    cm_held[q] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cm[q][k] = 0.f;
  }
  // Normalize.forward is (x - offset) / scale (norm.py:40-41); here the quotient is a multiplication by the
  // reciprocal taken once per wave (<= 1.5 ulp from the exact division, against the 1e-5 bar): the exact fp32
  // division costs a dozen VALU instructions per feature in a kernel whose epilogue is issue-bound
  float mel_off = 0.f, mel_inv = 1.f;
  if (MEL != 0 && p.offset) {
    mel_off = *p.offset;
    mel_inv = 1.0f / *p.scale;
  }

  // one frame: window, FFT, merge, store; `n6`/`n7` are the next frame's two new segments, already requested.
  // The Nyquist bin (a one-lane, exec-masked store the compiler cannot count on) is deferred to the top of the
  // next iteration, *before* that iteration's loads: the wait for the prefetched samples then has exactly
  // eight younger stores behind it and leaves all of them in flight.
  float2 nyq_pending = make_float2(0.f, 0.f);
  float2* nyq_dst = nullptr;
  auto flush_nyquist = [&]() {
    if (nyq_dst != nullptr && lane == 0) *nyq_dst = nyq_pending;
    if (WRITE_PHASE && nyq_dst != nullptr && lane == 0)
      p.phase[(nyq_dst - p.out)] = fast_atan2f(nyq_pending.y, nyq_pending.x);
  };
  auto frame_body = [&](const float2 (&fresh)[HS]) {
    // Wave priority.  Four waves share a SIMD, each somewhere else in its frame: the transform is one long chain of
    // dependent packed arithmetic between LDS exchanges, the stores and the epilogue are short bursts between waits.
    // Left to the default arbitration a wave in its epilogue takes issue slots from a wave in its transform, and every
    // frame's stores leave later than they could.  Transform 3 > stores 1 > epilogue 0 (same-box A/B of six shapes,
    // profiles/r04y_wave_priority.md): fused forward -1.6 % on a box whose memory is slow with mixed traffic, -4 % on
    // the others; features only -5.5 %; plain -2 %.  The inverse has no such phases (two shapes tried: 0 / +3 %).
    wave_priority<3>();
    float2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v2f w = (HYB & 2) ? win_regs[m] : lds_read_single(reinterpret_cast<const v2f*>(win) + lane + 64 * m);
      v[m] = make_float2(raw[m].x * w.x, raw[m].y * w.y);
    }
#pragma unroll
    for (int m = 0; m < 8 - HS; ++m) raw[m] = raw[m + HS];
#pragma unroll
    for (int k = 0; k < HS; ++k) raw[8 - HS + k] = fresh[k];
    float2 nyq;
    const int col = AL ? ((lane - rot) & 63) : lane;      // this lane ends up with bins col + 64 m
    {
      v2f z[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) z[m] = to_v(v[m]);
      if constexpr (AL && (HYB & 1)) {
        fft512<false>(z, tw_hyb, lds, lane, col);
        tw_hyb.col = col;
        rfft_merge_rot(z, tw_hyb, lane, rot, col, nyq);
      } else if constexpr (AL) {
        fft512<false>(z, tw_lds, lds, lane, col);
        const LdsTwiddles<false> tw_col = {tab, col};
        rfft_merge_rot(z, tw_col, lane, rot, col, nyq);
      } else if constexpr (TWLDS && (HYB & 1)) {
        fft512<false>(z, tw_hyb, lds, lane);
        rfft_merge(z, tw_hyb, lane, nyq);
      } else if (TWLDS) {
        fft512<false>(z, tw_lds, lds, lane);
        rfft_merge(z, tw_lds, lane, nyq);
      } else {
        fft512<false>(z, tw_regs, lds, lane);
        rfft_merge(z, tw_regs, lane, nyq);
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = to_f2(z[m]);
      if constexpr (AL) {
        wave_priority<1>();
        const bool lo = lane < rot;
        const v2f s0 = lo ? carry : z[0];
        if (head) {
          if (!lo) put(sp, s0);
          head = false;
        } else {
          put(sp, s0);
        }
#pragma unroll
        for (int j = 1; j < 8; ++j) put(sp + 64 * j, lo ? z[j - 1] : z[j]);
        carry = lo ? z[7] : (v2f){nyq.x, 0.f};
        if (rot == 63) {       // the frame ends exactly on a block boundary
          put(sp + 512, carry);
          sp += 576;
          rot = 0;
        } else {
          sp += 512;
          ++rot;
        }
      }
    }
    if (MEL != 2 && !AL) {
#pragma unroll
      for (int m = 0; m < 8; ++m) row[lane + 64 * m] = v[m];
    }
    if (WRITE_PHASE) {
#pragma unroll
      for (int m = 0; m < 8; ++m) prow[lane + 64 * m] = fast_atan2f(v[m].y, v[m].x);
      prow += F;
    }
    if (POLAR) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float ph = fast_atan2f(v[m].y, v[m].x);
        if (p.ph_offset) ph = (ph - ph_off) / ph_sc;
        prow[lane + 64 * m] = ph;
      }
      if (lane == 0) {
        float ph = fast_atan2f(nyq.y, nyq.x);
        if (p.ph_offset) ph = (ph - ph_off) / ph_sc;
        prow[512] = ph;
      }
      prow += phase_ld;
    }
    // channel-major register window of pass CMROW: take this frame's value, flush when the window ends on a 32-byte
    // boundary of the lane's own (.., N, T) row or at the end of the run
    auto cm_push = [&](auto cmsel, int f, float value) {
      constexpr int CMROW = decltype(cmsel)::value;
      if (f >= 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) cm[CMROW][k] = cm[CMROW][k + 1];
        cm[CMROW][7] = value;
      }
      // Every lane flushes its window when the eight frames it holds end on a 32-byte boundary of its own output row
      // ((.., N, T) rows start at arbitrary multiples of 4 bytes), so that a store covers whole 32-byte sectors; a
      // window cut by the frame index instead straddles two sectors on most rows and both are written partially
      // (TCC_EA0_WRREQ 16.2 M per launch, 4.8 M of them whole 64-byte requests).  At the end of the run: whatever it holds.
      ++cm_held[CMROW];
      if (f >= 0) {
        const long long e = ((long long)b * p.bank.n_filters + f) * p.T + t_cur + 1;   // one past frame t_cur
        if ((e & 7) == 0 || t_cur == t1 - 1) {
          float* dst = p.feat + e - 8;
          if (cm_held[CMROW] >= 8) {
            if ((e & 3) == 0) {
              reinterpret_cast<float4*>(dst)[0] = make_float4(cm[CMROW][0], cm[CMROW][1], cm[CMROW][2], cm[CMROW][3]);
              reinterpret_cast<float4*>(dst)[1] = make_float4(cm[CMROW][4], cm[CMROW][5], cm[CMROW][6], cm[CMROW][7]);
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) dst[k] = cm[CMROW][k];
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (k >= 8 - cm_held[CMROW]) dst[k] = cm[CMROW][k];
          }
          cm_held[CMROW] = 0;
        }
      }
    };
    wave_priority<0>();
    if constexpr (MEL != 0 && FQ0 > 0) {
      static_assert(SP == 2 && (CMBUF == 0 || CMBUF == 2) && !POLAR, "fixed-length epilogue: two passes, no phase rows");
      float* absrow = reinterpret_cast<float*>(lds);
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; ++m)
        absrow[col + 64 * m] = FP2 ? fmaf(v[m].x, v[m].x, v[m].y * v[m].y)
                                   : __builtin_amdgcn_sqrtf(fmaf(v[m].x, v[m].x, v[m].y * v[m].y));
      // bin 512.  Entries 513 .. 639 are whatever the FFT left in the slab: a walk that runs past its band
      // multiplies them by zero weights (finite leftovers: the frame's own intermediate values; a frame
      // that holds inf / NaN yields NaN features either way)
      if (col == 0) absrow[512] = FP2 ? nyq.x * nyq.x : fabsf(nyq.x);
      wave_lds_sync();
      float sum0, sum1;
      band_walk_fixed<FQ0, FQ1>(absrow, sp_start[0], sp_start[1], wlds, lane, sum0, sum1);
      // FC: the contrast as a compile-time constant (1 = log1p, the headline; 2 = log of the clamped value: the log-mel of
      // BASELINE configs[3]); FP2: |X|^2 instead of |X|
      float f0v = band_contrast_fast(sum0, FC, p.eps);
      float f1v = band_contrast_fast(sum1, FC, p.eps);
      f0v = (f0v - mel_off) * mel_inv;      // no Normalize: offset 0, reciprocal 1 -- the identity, bit for bit
      f1v = (f1v - mel_off) * mel_inv;
      if constexpr (CMBUF == 2) {
        cm_push(std::integral_constant<int, 0>(), sp_f[0], f0v);
        cm_push(std::integral_constant<int, 1>(), sp_f[1], f1v);
      } else {
        if (sp_f[0] >= 0) frow[sp_f[0]] = f0v;
        if (sp_f[1] >= 0) frow[sp_f[1]] = f1v;
      }
      wave_lds_sync();
      frow += feat_ld;
      ++t_cur;
    } else if constexpr (MEL != 0 && FQP != 0) {
      static_assert(SP == 0 && CMBUF == 0 && !POLAR && !FP2 && FC == 1, "packed fixed-length epilogue: row-major log1p(|X| bank) features");
      float* absrow = reinterpret_cast<float*>(lds);
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; ++m) absrow[col + 64 * m] = __builtin_amdgcn_sqrtf(fmaf(v[m].x, v[m].x, v[m].y * v[m].y));
      // bin 512; entries 513 .. 639 are finite leftovers of this frame's exchanges under zero weights (fft512.h pins that)
      if (col == 0) absrow[512] = fabsf(nyq.x);
      wave_lds_sync();
      const int4* pk4 = reinterpret_cast<const int4*>(lane_tab + 2 * 64 * kNP + lane * kPkStride);
      int a_off[kNP];
#pragma unroll
      for (int g = 0; g < kPkStride / 4; ++g) {
        const int4 d = pk4[g];
        if (4 * g + 0 < kNP) a_off[4 * g + 0] = d.x & 0xfff;
        if (4 * g + 1 < kNP) a_off[4 * g + 1] = d.y & 0xfff;
        if (4 * g + 2 < kNP) a_off[4 * g + 2] = d.z & 0xfff;
        if (4 * g + 3 < kNP) a_off[4 * g + 3] = d.w & 0xfff;
      }
      float sums[kNP];
      band_walk_packed<FQP, kNP>(absrow, a_off, wlds, lane, sums);
      // The lanes of a pass hold filters scattered over the whole row (passes are cut by band length, lanes placed for
      // conflict-free magnitude reads): stored straight from the lanes, each pass is a 64-lane scatter of 4-byte stores
      // over ~30 different 64-byte segments -- ~270 write requests per frame for 2 KB of features, more than the 4 KB
      // spectrum row costs.  The row is put in order through LDS instead (the magnitudes are dead once every walk is
      // done), BEHIND the fc floats carried from the frame before, and whole 1-KB aligned blocks of the feature stream
      // leave: 32 write requests per frame (two 1-KB stores at the row's own 4-byte-aligned base were 51: a lane's 16
      // bytes straddle a 64-byte boundary on three rows of four) -- the spectrum-storing form is bound by exactly that
      // count (profiles/r04_default_bank_513.md).
      wave_lds_sync();
      float* fbuf = reinterpret_cast<float*>(lds) - 2 * kFeatCarry;      // [0, 256): carried floats; [256, ..): the FFT slab
#pragma unroll
      for (int g = 0; g < kPkStride / 4; ++g) {
        const int4 d = pk4[g];        // read again rather than held across the walk
        const int dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (4 * g + k < kNP) {
            const float fv = (band_contrast_fast(sums[4 * g + k], 1, p.eps) - mel_off) * mel_inv;
            const int f_off = (int)((unsigned)dd[k] >> 12);
            if (f_off != 0xfff) *reinterpret_cast<float*>(reinterpret_cast<char*>(fbuf + fc) + f_off) = fv;
          }
        }
      }
      wave_lds_sync();
      {
        const int n = fc + F;             // floats in hand: two or three whole blocks and a remainder
        const int nblk = n >> 8;
        const float4* fb4 = reinterpret_cast<const float4*>(fbuf);
        if (fhead > 0) {                  // the run's first block: its leading fhead floats are another run's
          const float4 v = fb4[lane];
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (4 * lane + c >= fhead) fdst[4 * lane + c] = vv[c];
        } else {
          *reinterpret_cast<float4*>(fdst + 4 * lane) = fb4[lane];
        }
        fhead = 0;
        *reinterpret_cast<float4*>(fdst + 256 + 4 * lane) = fb4[64 + lane];
        if (nblk == 3) *reinterpret_cast<float4*>(fdst + 512 + 4 * lane) = fb4[128 + lane];
        const float4 rem = fb4[64 * nblk + lane];       // the remainder moves to the front (junk behind it is overwritten)
        wave_lds_sync();
        reinterpret_cast<float4*>(fbuf)[lane] = rem;
        fc = n & 255;
        fdst += 256 * nblk;
      }
      wave_lds_sync();
      frow += feat_ld;
      ++t_cur;
    } else if (MEL != 0) {
      // |X| (or |X|^2) of this frame into the wave's LDS slab (free again after the FFT), then every lane
      // gathers the bands of its filters: sum_k |X[k]| w[k][n] over the band only (the rest of the column is 0)
      float* absrow = reinterpret_cast<float*>(lds);
      wave_lds_sync();
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float s2 = fmaf(v[m].x, v[m].x, v[m].y * v[m].y);
        absrow[col + 64 * m] = p.power2 ? s2 : __builtin_amdgcn_sqrtf(s2);
      }
      // bin 512, then zeros: a walk may run past its band (zero weights there), up to entry 639
      absrow[512 + col] = (col == 0) ? (p.power2 ? nyq.x * nyq.x : fabsf(nyq.x)) : 0.0f;
      absrow[576 + lane] = 0.0f;
      wave_lds_sync();
      const float4* w = reinterpret_cast<const float4*>(wlds) + lane;
      // one pass: this lane's filter of pass q summed over its band, contrast / normalise, store (or park in
      // the channel-major register window `cmrow`, a compile-time row of `cm`)
      auto one_pass = [&](int q, auto cmsel) {
        constexpr int CMROW = decltype(cmsel)::value;     // -1: no register window
        const int f = SP > 0 ? sp_f[q] : lane_tab[(p.bank.n_passes + q) * 64 + lane];
        // one ds_read_b128 of magnitudes and one of weights per four multiply-adds
        const float4* a = reinterpret_cast<const float4*>(absrow + (SP > 0 ? sp_start[q] : lane_tab[q * 64 + lane]));
        // two running sums (even / odd bins of each pair) on packed multiply-adds, joined at the end
        v2f acc2 = {0.f, 0.f};
        const int quads = SP > 0 ? sp_quads[q] : p.bank.pass_len[q] >> 2;       // wave-uniform; shorter bands multiply zeros
        int j = 0;
        // four steps' reads issued before the first multiply-add: one LDS round trip per four steps, not per step
        for (; j + 4 <= quads; j += 4) {
          float4 av[4], wv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = a[j + u];
            wv[u] = w[(j + u) * 64];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc2 = __builtin_elementwise_fma((v2f){av[u].x, av[u].y}, (v2f){wv[u].x, wv[u].y}, acc2);
            acc2 = __builtin_elementwise_fma((v2f){av[u].z, av[u].w}, (v2f){wv[u].z, wv[u].w}, acc2);
          }
        }
        for (; j < quads; ++j) {
          const float4 av = a[j], wv = w[j * 64];
          acc2 = __builtin_elementwise_fma((v2f){av.x, av.y}, (v2f){wv.x, wv.y}, acc2);
          acc2 = __builtin_elementwise_fma((v2f){av.z, av.w}, (v2f){wv.z, wv.w}, acc2);
        }
        float acc = acc2.x + acc2.y;
        w += quads * 64;
        if (f >= 0) {
          acc = fwd_contrast(acc, p.contrast, p.eps);
          if (p.offset) acc = (acc - mel_off) * mel_inv;
          if constexpr (CMROW >= 0) {
            // parked in the register window below
          } else if (p.feat_channel_major) {
            p.feat[((long long)b * p.bank.n_filters + f) * p.T + t_cur] = acc;
          } else {
            frow[f] = acc;
          }
        }
        if constexpr (CMROW >= 0) cm_push(cmsel, f, acc);
      };
      if constexpr (CMBUF >= 1) {            // the launcher picks CMBUF == n_passes (1 or 2)
        one_pass(0, std::integral_constant<int, 0>());
        if constexpr (CMBUF >= 2) one_pass(1, std::integral_constant<int, 1>());
      } else if constexpr (SP >= 1) {        // likewise SP == n_passes
        one_pass(0, std::integral_constant<int, -1>());
        if constexpr (SP >= 2) one_pass(1, std::integral_constant<int, -1>());
      } else {
        for (int q = 0; q < p.bank.n_passes; ++q) one_pass(q, std::integral_constant<int, -1>());
      }
      wave_lds_sync();
      frow += feat_ld;
      ++t_cur;
    }
    nyq_pending = nyq;
    nyq_dst = (MEL != 2 && !AL) ? row + 512 : nullptr;
    row += F;
  };

  // Frames whose successor's new samples [H t + 512, H (t + 1) + 512) lie inside the clip take the
  // branch-free loop: HS unconditional 8-byte loads issued *before* this frame's stores, so the
  // compiler's vmcnt accounting lets the stores stay in flight across iterations.
  long long t = t0;
  long long t_fast_end = (L >= 512) ? (L - 512) / H : 0;         // first t whose successor needs reflection
  // The last frame of a run whose successor (another run's first frame) lies inside the clip fetches that frame's new
  // samples like any other and drops them: 1 KB read again by the neighbour (L2), against a trip through the generic loop
  // below -- per run, which is what makes short runs expensive.  Only the clip's last frames take the generic loop.
  if (t_fast_end > t1) t_fast_end = t1;
  if (!clip_aligned) t_fast_end = t0;                  // odd-length clips: generic loop only
  if (t < t_fast_end) {
    // segment 8 - HS of frame t+1 starts at original sample H (t+1) - 512 + 128 (8 - HS) = H t + 512
    const float2* nsrc = reinterpret_cast<const float2*>(clip + t * H + 512) + lane;
    for (; t < t_fast_end; ++t) {
      flush_nyquist();
      float2 fresh[HS];
#pragma unroll
      for (int k = 0; k < HS; ++k) fresh[k] = nsrc[64 * k];
      nsrc += H / 2;
      frame_body(fresh);
    }
  }
  for (; t < t1; ++t) {
    flush_nyquist();
    float2 fresh[HS];
#pragma unroll
    for (int k = 0; k < HS; ++k) fresh[k] = make_float2(0.f, 0.f);
    if (t + 1 < t1) {
#pragma unroll
      for (int k = 0; k < HS; ++k) {
        const long long i0 = t * H + 512 + 128 * k;
        fresh[k] = load_pair(clip, L, i0, i0 >= 0 && i0 + 128 <= L, clip_aligned, lane2);
      }
    }
    frame_body(fresh);
  }
  flush_nyquist();
  if constexpr (AL) {
    if (lane < rot) put(sp, carry);      // the run's last block: the next run (or clip) owns the rest of it
  }
  if constexpr (FQP != 0) {              // likewise the feature stream's last, partial block
    wave_lds_sync();
    const float* fbuf = reinterpret_cast<const float*>(lds) - 2 * kFeatCarry;
    const float4 v = reinterpret_cast<const float4*>(fbuf)[lane];
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (4 * lane + c < fc && 4 * lane + c >= fhead) fdst[4 * lane + c] = vv[c];
    wave_lds_sync();
  }
  };   // do_run

  if constexpr (!PW) {
    do_run((long long)blockIdx.x * FWD_WAVES + wave, true);
  } else {
    for (int it = 0;; ++it) {
      const unsigned tile = __builtin_amdgcn_readfirstlane(s_tile[it & 1]);
      if (tile >= p.n_tiles) break;
      unsigned nxt = 0;
      if (threadIdx.x == 0) nxt = atomicAdd(p.tile_ctr, 1u);       // consumed after this tile's run
      do_run((long long)tile * FWD_WAVES + wave, false);
      if (threadIdx.x == 0) s_tile[(it + 1) & 1] = nxt;
      __syncthreads();
    }
    // the last workgroup to leave re-arms the counter pair for the next launch that is given this slot
    if (threadIdx.x == 0) {
      if (atomicAdd(p.tile_ctr + 1, 1u) == gridDim.x - 1) {
        atomicExch(p.tile_ctr, 0u);
        atomicExch(p.tile_ctr + 1, 0u);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------
enum { IN_COMPLEX = 0, IN_POLAR = 1, IN_GL = 2 };
enum { OUT_OLA = 0, OUT_FRAMES = 1 };

struct InvParams {
  const float2* X;      // (B*T, 513) complex64           (IN_COMPLEX)
  const float* mag;     // (B*T, 513)                      (IN_POLAR)
  const float* phase;   // (B*T, 513)                      (IN_POLAR)
  const float2* tprev;  // (B*T, 513) complex64           (IN_GL: X = the rebuilt spectrum, mag = the target magnitudes)
  float gl_mom;         // IN_GL: momentum / (1 + momentum)
  const float* window;  // N synthesis window samples
  const float* env;     // OUT_OLA: 16 x hop table, env[mask][r] = sum of window^2 over the frames in mask
  const float2* tw;
  float* y;             // OUT_OLA: (B, hop*(T-1));  OUT_FRAMES: (B*T, 1024)
  long long B, T;
  long long runs_per_clip;  // OUT_OLA
  long long slots_per_run;  // OUT_OLA
  long long total_frames, frames_per_block;  // OUT_FRAMES
};

// cos/sin of an unwrapped phase that may be ~1e5 rad (PGHI): reduce in fp64 to revolutions in
// [-0.5, 0.5] (exact to ~1e-16), then the hardware sin/cos (v_sin_f32 takes revolutions; abs error ~1e-6).
__device__ __forceinline__ void sincos_big(float phase, float& s, float& c) {
  double t = (double)phase * 0.15915494309189533577;  // 1 / (2 pi)
  t -= rint(t);
  const float r = (float)t;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

// One frame's spectrum as it sits in HBM, loaded ahead of use and converted when consumed:
//   IN_COMPLEX: d[m] = X[lane + 64 m],             ny0 = Re X[512]
//   IN_POLAR:   d[m] = (mag, phase)[lane + 64 m],  (ny0, ny1) = (mag, phase)[512]
// (the polar form keeps mag and phase in separate registers: each is the target of its own dword load)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int IN_MODE>
struct RawFrame;
template <>
struct RawFrame<IN_COMPLEX> {
  f32x2 d[8];
  float ny0;
};
template <>
struct RawFrame<IN_POLAR> {
  float a[8], ph[8];
  float ny0, ny1;
};
// IN_GL: the Griffin-Lim phase update (torchaudio.functional.griffinlim as called at reference stft.py:174-178)
// taken at load time -- X = mag * normalise(rebuilt - m' tprev) never exists in HBM
template <>
struct RawFrame<IN_GL> {
  f32x2 r[8], t[8];
  float a[8];
  f32x2 rny, tny;
  float any;
};

__device__ __forceinline__ void load_raw(const InvParams& p, long long f, int lane, RawFrame<IN_COMPLEX>& q) {
  const f32x2* row = reinterpret_cast<const f32x2*>(p.X + f * F);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
#if AT_ISTFT_NTLOAD
    q.d[m] = __builtin_nontemporal_load(row + lane + 64 * m);
#else
    q.d[m] = row[lane + 64 * m];
#endif
  }
  // broadcast load; only lane 0 uses it.  A 4-byte load of the real part alone: as `row[512].x` it was an 8-byte
  // load whose dead upper register the allocator handed to the next instruction at once -- a write-after-write
  // hazard on a load just issued, i.e. `s_waitcnt vmcnt(0)` in the steady-state loop, draining both frames of lookahead.
  q.ny0 = reinterpret_cast<const float*>(row + 512)[0];
}
__device__ __forceinline__ void load_raw(const InvParams& p, long long f, int lane, RawFrame<IN_POLAR>& q) {
  const float* mrow = p.mag + f * F;
  const float* prow = p.phase + f * F;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    q.a[m] = mrow[lane + 64 * m];
    q.ph[m] = prow[lane + 64 * m];
  }
  q.ny0 = mrow[512];
  q.ny1 = prow[512];
}

__device__ __forceinline__ void load_raw(const InvParams& p, long long f, int lane, RawFrame<IN_GL>& q) {
  const f32x2* rrow = reinterpret_cast<const f32x2*>(p.X + f * F);
  const f32x2* trow = reinterpret_cast<const f32x2*>(p.tprev + f * F);
  const float* mrow = p.mag + f * F;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    q.r[m] = rrow[lane + 64 * m];
    q.t[m] = trow[lane + 64 * m];
    q.a[m] = mrow[lane + 64 * m];
  }
  q.rny = rrow[512];
  q.tny = trow[512];
  q.any = mrow[512];
}

__device__ __forceinline__ float2 gl_update(f32x2 r, f32x2 t, float mag, float mom) {
  float ax = r.x, ay = r.y;
  ax -= mom * t.x;
  ay -= mom * t.y;
  // |a| + 1e-16 and the two quotients through the hardware sqrt / rcp (1 ulp each): this runs 513 times per frame
  // in front of the FFT, where libm's hypotf and two IEEE divisions (~50 instructions) made the kernel VALU-bound.
  // Audio-scale spectra are nowhere near the range where hypot's rescaling matters.
  const float d = __builtin_amdgcn_sqrtf(fmaf(ax, ax, ay * ay)) + 1e-16f;
  const float s = mag * __builtin_amdgcn_rcpf(d);
  return make_float2(s * ax, s * ay);
}

__device__ __forceinline__ void raw_to_spectrum(const RawFrame<IN_COMPLEX>& q, float2 (&v)[8], float& nyq_re) {
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = make_float2(q.d[m].x, q.d[m].y);
  nyq_re = q.ny0;
}
__device__ __forceinline__ void raw_to_spectrum(const RawFrame<IN_POLAR>& q, float2 (&v)[8], float& nyq_re) {
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    float sn, cs;
    sincos_big(q.ph[m], sn, cs);
    v[m] = make_float2(q.a[m] * cs, q.a[m] * sn);
  }
  float sn, cs;
  sincos_big(q.ny1, sn, cs);
  nyq_re = q.ny0 * cs;
}

__device__ __forceinline__ void raw_to_spectrum(const RawFrame<IN_GL>& q, float2 (&v)[8], float& nyq_re, float mom) {
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = gl_update(q.r[m], q.t[m], q.a[m], mom);
  nyq_re = gl_update(q.rny, q.tny, q.any, mom).x;
}

template <int IN_MODE>
__device__ __forceinline__ void load_spectrum(const InvParams& p, long long f, int lane, float2 (&v)[8], float& nyq_re) {
  RawFrame<IN_MODE> q;
  load_raw(p, f, lane, q);
  raw_to_spectrum(q, v, nyq_re);
}

// the synthesis window with the transform's 1/N folded in (a power of two: the products round the same)
__device__ __forceinline__ float2 scaled_window(const float* window, int i) {
  const float2 w = reinterpret_cast<const float2*>(window)[i];
  return make_float2(w.x * (1.0f / 1024.0f), w.y * (1.0f / 1024.0f));
}

// one frame: spectrum -> time samples z[m] = (x[2n], x[2n+1]), n = lane + 64 m, BEFORE the synthesis window.  The
// overlap-add kernels take the window inside their accumulation, acc = fma(z, w, acc), spelled out: left to the
// compiler's contraction the same sum came out as fma in one kernel and as round(z w) + acc in another (whose pieces
// travel through LDS), and a clip's bits must not depend on which kernel its batch size selects.
template <typename TW>
__device__ __forceinline__ void synth_frame_nowin(const float2 (&v)[8], float nyq_re, const TW& tw, float2* lds, int lane,
                                                  v2f (&z)[8]) {
#pragma unroll
  for (int m = 0; m < 8; ++m) z[m] = to_v(v[m]);
  irfft_split(z, tw, lane, nyq_re);
  fft512<true>(z, tw, lds, lane);
}
__device__ __forceinline__ v2f ola_window(const float2* win, int lane, int m) {
  return lds_read_single(reinterpret_cast<const v2f*>(win) + lane + 64 * m);   // window / 1024, workgroup-shared LDS copy
}

// one frame: spectrum -> windowed time samples y[m] = (x[2n], x[2n+1]) * w, n = lane + 64 m
template <typename TW>
__device__ __forceinline__ void synth_frame(float2 (&v)[8], float nyq_re, const TW& tw, const float2* win,
                                            float2* lds, int lane) {
  v2f z[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) z[m] = to_v(v[m]);
  irfft_split(z, tw, lane, nyq_re);
  fft512<true>(z, tw, lds, lane);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const v2f w = lds_read_single(reinterpret_cast<const v2f*>(win) + lane + 64 * m);   // window / 1024, workgroup-shared LDS copy
    v[m] = to_f2(z[m] * w);
  }
}

// K3: irFFT + window + overlap-add + envelope division + centre trim, for hop = 128 HS (HS = 1, 2, 4: N/8, N/4 --
// the reference's default --, N/2).  A lane's eight registers are eight 128-sample slots of the padded signal, a hop
// is HS of them, R = 8 / HS frames overlap and the centre trim is C = 4 / HS hops.
// A wave produces output hop-slots [j0, j1) of one clip, streaming over frames j0+C-(R-1) .. j1-1+C with the R
// overlapping frames' partial sums in registers.  Spectra are requested two frames ahead (8 KB per wave in
// flight); the steady-state loop is branch-free so that the compiler's vmcnt accounting leaves those loads
// and the previous outputs' stores in flight across iterations (a conditional load anywhere in the loop
// collapses every wait to vmcnt(0)).
template <int IN_MODE, int DEPTH, bool TWLDS, int HS = 2>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, IN_MODE == IN_GL ? 2 : (TWLDS ? 5 : 4) - DEPTH) void istft1024_ola_kernel(InvParams p) {
  constexpr int R = 8 / HS;          // overlapping frames
  constexpr int C = 4 / HS;          // centre trim in hops
  constexpr int H = 128 * HS;        // hop in samples
  constexpr int kFull = (1 << R) - 1;
  __shared__ float2 lds_all[WAVES_PER_BLOCK * kFftLdsFloat2PerWave + 512 + (TWLDS ? kTwiddleCount : 0)];
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; said explicitly, or everything derived from it (clip, run bounds, the frame
  // counter of the main loop, the column rotation) lives in vector registers and the loop control runs on the VALU
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* win = lds_all + WAVES_PER_BLOCK * kFftLdsFloat2PerWave;
  float2* twtab = win + 512;
  for (int i = threadIdx.x; i < 512; i += 64 * WAVES_PER_BLOCK) win[i] = scaled_window(p.window, i);
  if (TWLDS)
    for (int i = threadIdx.x; i < kTwiddleCount; i += 64 * WAVES_PER_BLOCK) twtab[i] = p.tw[i];
  __syncthreads();

  const long long run = (long long)blockIdx.x * WAVES_PER_BLOCK + wave;
  const long long b = run / p.runs_per_clip;
  if (b >= p.B) return;
  const long long r = run - b * p.runs_per_clip;
  const long long nslots = p.T - 1;
  const long long j0 = r * p.slots_per_run;
  long long j1 = j0 + p.slots_per_run;
  if (j1 > nslots) j1 = nslots;
  if (j0 >= j1) return;

  Twiddles tw_regs;
  if (!TWLDS) load_twiddles<true>(tw_regs, p.tw, lane);
  const LdsTwiddles<true> tw_lds = {twtab, lane};
  auto synth = [&](const float2 (&v)[8], float nyq, v2f (&z)[8]) {
    if (TWLDS) synth_frame_nowin(v, nyq, tw_lds, lds, lane, z);
    else synth_frame_nowin(v, nyq, tw_regs, lds, lane, z);
  };

  v2f acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = (v2f){0.f, 0.f};

  const long long fbase = b * p.T;
  float* yclip = p.y + b * (H * nslots);
  long long t = (j0 + C - (R - 1) > 0) ? j0 + C - (R - 1) : 0;
  const long long t_last = j1 - 1 + C;                              // inclusive; frames >= T contribute nothing
  const long long t_have = (t_last < p.T - 1) ? t_last : p.T - 1;   // last frame this run reads

  RawFrame<IN_MODE> q0 = {}, q1 = {};   // frames t and t + 1
  if (t <= t_have) load_raw(p, fbase + t, lane, q0);
  if (t + 1 <= t_have) load_raw(p, fbase + t + 1, lane, q1);

  // accumulators start aligned with frame t: acc[m] covers padded samples t*H + 2*(lane+64m)
  auto consume = [&](const RawFrame<IN_MODE>& q) {
    float2 v[8];
    float nyq;
    if constexpr (IN_MODE == IN_GL) raw_to_spectrum(q, v, nyq, p.gl_mom);
    else raw_to_spectrum(q, v, nyq);
    v2f z[8];
    synth(v, nyq, z);
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = __builtin_elementwise_fma(z[m], ola_window(win, lane, m), acc[m]);
  };
  // finished hops leave through non-temporal stores: written once, read by nobody here (the pattern-only copy kernel
  // of tools/ubench/stream_pattern2.hip gains 3-4 % from them: 0.717 -> 0.689 ms)
  auto put_y = [&](float2* dst, float2 val) {
#if AT_ISTFT_NT
    __builtin_nontemporal_store((v2f){val.x, val.y}, reinterpret_cast<v2f*>(dst));
#else
    *dst = val;
#endif
  };
  static_assert(sizeof(v2f) == sizeof(float2), "");
  auto emit = [&](long long j, const float2* env) {
    float2* dst = reinterpret_cast<float2*>(yclip + j * H);
#pragma unroll
    for (int k = 0; k < HS; ++k) {
      const float2 e = env[lane + 64 * k];
      put_y(dst + lane + 64 * k, make_float2(acc[k].x / e.x, acc[k].y / e.y));
    }
  };
  // steady state: the envelope of a fully overlapped hop is the same for every frame, so its reciprocal is taken
  // once per wave (an fp32 division is ~10 instructions, four of them per frame); at most one ulp from acc / e
  float2 rcp[HS];
  auto emit_fast = [&](long long j) {
    float2* dst = reinterpret_cast<float2*>(yclip + j * H);
#pragma unroll
    for (int k = 0; k < HS; ++k) put_y(dst + lane + 64 * k, make_float2(acc[k].x * rcp[k].x, acc[k].y * rcp[k].y));
  };
  auto advance = [&]() {   // one hop = HS register slots
#pragma unroll
    for (int m = 0; m < 8 - HS; ++m) acc[m] = acc[m + HS];
#pragma unroll
    for (int m = 8 - HS; m < 8; ++m) acc[m] = (v2f){0.f, 0.f};
  };
  // reciprocal of the full-overlap envelope, shared by the steady-state loop and the edge steps: a hop's value
  // must not depend on which of the two emitted it, i.e. on how the launch cut the clip into runs
  {
    const float2* env_full = reinterpret_cast<const float2*>(p.env + kFull * H);
#pragma unroll
    for (int k = 0; k < HS; ++k) {
      const float2 e = env_full[lane + 64 * k];
      rcp[k] = make_float2(1.0f / e.x, 1.0f / e.y);
    }
  }
  // edges of the run / of the clip: any frame or slot may be missing
  auto generic_step = [&]() {
    if (t <= t_have) {
      const RawFrame<IN_MODE> cur = q0;
      q0 = q1;
      if (t + 2 <= t_have) load_raw(p, fbase + t + 2, lane, q1);
      consume(cur);
    }
    const long long j = t - C;  // padded hop t is complete -> output slot j, fed by frames t-(R-1) .. t
    if (j >= j0 && j < j1) {
      int mask = 0;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const long long tt = t - (R - 1) + q;
        if (tt >= 0 && tt < p.T) mask |= 1 << q;
      }
      if (mask == kFull) emit_fast(j);
      else emit(j, reinterpret_cast<const float2*>(p.env + mask * H));
    }
    advance();
    ++t;
  };

  // steady state: frames t .. t+2 exist, slot t-C belongs to this run and has all R contributors
  long long fast_begin = j0 + C > R - 1 ? j0 + C : R - 1;
  long long fast_end = (j1 - 1 + C < t_have - 2) ? j1 - 1 + C : t_have - 2;   // inclusive
  while (t < fast_begin && t <= t_last) generic_step();
  if (t <= fast_end) {
    if (DEPTH == 2) {
      // two frames in flight, two frames per trip with q0 / q1 swapping roles
      auto fast_step = [&](RawFrame<IN_MODE>& q) {
        const RawFrame<IN_MODE> cur = q;
        load_raw(p, fbase + t + 2, lane, q);
        consume(cur);
        emit_fast(t - C);
        advance();
        ++t;
      };
      while (t + 1 <= fast_end) {
        fast_step(q0);
        fast_step(q1);
      }
    } else {
      // one frame in flight (q0 = frame t); q1 is re-requested by the generic tail
      q1 = {};
      for (; t <= fast_end; ++t) {
        const RawFrame<IN_MODE> cur = q0;
        load_raw(p, fbase + t + 1, lane, q0);
        consume(cur);
        emit_fast(t - C);
        advance();
      }
      q1 = {};
      if (t + 1 <= t_have) load_raw(p, fbase + t + 1, lane, q1);
    }
  }
  while (t <= t_last) generic_step();
}

// K3 for full batches (round 5): the same transform, the same additions in the same order, cut for the memory system.
// The kernel above gives each wave ONE long run of a clip (173 hops at 1024 clips) and pays three warm-up frames per run;
// the chip then reads 2048 fronts a run length apart.  The inverse's access pattern runs 6-8 % faster when the
// workgroups of a launch walk the streams in address order (tools/ubench/stream_pattern3.hip, "inv D": 0.61 against
// 0.65 ms), but at 16-frame runs the three warm-up frames would be 19 % more transforms.  Here a WORKGROUP owns a tile of
// consecutive frames of one clip and the overlap state crosses the cuts between its waves through LDS instead of being
// recomputed:
//   * wave w transforms ONLY its own frames [ta, tb) and emits the hops [ta + 3, tb + 3) (hop p = frames p-3 .. p);
//   * the first three frames of a wave also feed the last three hops of the wave before it: their hop-sized pieces
//     (3 + 2 + 1 = six 1-KB pieces) are parked in LDS UNSUMMED;
//   * after ONE workgroup barrier wave w adds the pieces of wave w + 1 to its three open hops one frame at a time -- the
//     additions happen in frame order, ((F[p-3] + F[p-2]) + F[p-1]) + F[p] exactly as in the long-run kernel, so the bits do
//     not depend on where the cuts are (or on which of the two kernels ran);
//   * the tile's last wave has no successor in the workgroup and transforms the next tile's first three frames itself
//     (3 extra transforms per tile: 1.7 % at the 173-frame tiles of a 690-frame clip; it is given three frames fewer so that the
//     waves finish together).
// No wave ever waits for another wave's transforms: by the time a wave has finished its own frames its successor parked
// its pieces long ago.  Tiles are dispatched in address order (tile = blockIdx.x).
template <int IN_MODE, int NW, int OCC>
__global__ __launch_bounds__(64 * NW, OCC) void istft1024_tile_kernel(InvParams p) {
  constexpr int HS = 2, R = 4, H = 256;
  constexpr int kFull = (1 << R) - 1;
  static_assert(IN_MODE != IN_GL, "the Griffin-Lim form keeps the long-run kernel");
  __shared__ float2 lds_all[NW * kFftLdsFloat2PerWave + 512];
  __shared__ float4 park_all[NW * 6 * 64];          // wave w: six hop-sized pieces of its first three frames
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* win = lds_all + NW * kFftLdsFloat2PerWave;

  // the tile and this wave's share of it (everything wave-uniform)
  const long long tile = blockIdx.x;
  const long long b = tile / p.runs_per_clip;             // runs_per_clip: tiles per clip
  const long long k = tile - b * p.runs_per_clip;
  const long long n = p.slots_per_run;                     // frames per wave (the last wave of a full tile: n - 3)
  const long long T = p.T;
  const long long tile0 = k * p.frames_per_block;          // frames_per_block: frames per tile = (NW - 1) n + n - 3
  long long tile1 = tile0 + p.frames_per_block;
  if (tile1 > T) tile1 = T;
  auto share = [&](int w, long long& a, long long& e) {
    a = tile0 + w * n;
    if (a > tile1) a = tile1;
    e = (w == NW - 1) ? tile1 : a + n;
    if (e > tile1) e = tile1;
  };
  long long ta, tb;
  share(wave, ta, tb);
  bool self_cool = true;                                    // this wave closes its last three hops itself
  if (wave < NW - 1) {
    long long a2, e2;
    share(wave + 1, a2, e2);
    self_cool = e2 - a2 < 3;
  }
  long long t_stop = (self_cool && tb > ta) ? tb + 3 : tb;  // frames this wave transforms: [ta, t_stop)
  if (t_stop > T) t_stop = T;
  const bool parks = wave > 0 && ta > 0;                    // somebody in this workgroup may read the pieces
  const long long pe0 = ta == 0 ? 2 : ta + 3;               // first padded hop this wave emits (hops 0, 1 are trimmed)
  const long long pe1 = tb + 3;                             // one past the last (hop T is the clip's last)

  const long long fbase = b * T;
  float* yclip = p.y + b * (H * (T - 1));
  long long t = ta;
  RawFrame<IN_MODE> q0 = {}, q1 = {};                       // frames t and t + 1, requested before the tables are staged
  if (t < t_stop) load_raw(p, fbase + t, lane, q0);
  if (t + 1 < t_stop) load_raw(p, fbase + t + 1, lane, q1);

  for (int i = threadIdx.x; i < 512; i += 64 * NW) win[i] = scaled_window(p.window, i);
  Twiddles tw;
  load_twiddles<true>(tw, p.tw, lane);
  __syncthreads();

  v2f acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = (v2f){0.f, 0.f};
  // z: the frame's samples before the window (what is parked); the window enters in the accumulation, one fma per sample
  auto consume = [&](const RawFrame<IN_MODE>& q, v2f (&z)[8]) {
    float2 v[8];
    float nyq;
    raw_to_spectrum(q, v, nyq);
    synth_frame_nowin(v, nyq, tw, lds, lane, z);
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = __builtin_elementwise_fma(z[m], ola_window(win, lane, m), acc[m]);
  };
  auto put_y = [&](float2* dst, float2 val) {
    __builtin_nontemporal_store((v2f){val.x, val.y}, reinterpret_cast<v2f*>(dst));
  };
  float2 rcp[HS];
  {
    const float2* env_full = reinterpret_cast<const float2*>(p.env + kFull * H);
#pragma unroll
    for (int kk = 0; kk < HS; ++kk) {
      const float2 e = env_full[lane + 64 * kk];
      rcp[kk] = make_float2(1.0f / e.x, 1.0f / e.y);
    }
  }
  // hop `hp` from accumulator slots a0, a1: the same two forms as the long-run kernel (reciprocal of the full envelope /
  // division by the partial one at the clip's two ends)
  auto emit_regs = [&](long long hp, v2f a0, v2f a1) {
    float2* dst = reinterpret_cast<float2*>(yclip + (hp - 2) * H);
    if (hp >= 3 && hp < T) {
      put_y(dst + lane, make_float2(a0.x * rcp[0].x, a0.y * rcp[0].y));
      put_y(dst + lane + 64, make_float2(a1.x * rcp[1].x, a1.y * rcp[1].y));
    } else {
      int mask = 0;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const long long tt = hp - (R - 1) + q;
        if (tt >= 0 && tt < T) mask |= 1 << q;
      }
      const float2* env = reinterpret_cast<const float2*>(p.env + mask * H);
      const float2 e0 = env[lane], e1 = env[lane + 64];
      put_y(dst + lane, make_float2(a0.x / e0.x, a0.y / e0.y));
      put_y(dst + lane + 64, make_float2(a1.x / e1.x, a1.y / e1.y));
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int m = 0; m < 8 - HS; ++m) acc[m] = acc[m + HS];
#pragma unroll
    for (int m = 8 - HS; m < 8; ++m) acc[m] = (v2f){0.f, 0.f};
  };
  float4* park = park_all + wave * 6 * 64 + lane;
  // any frame: the run's head (pieces parked, hops not yet this wave's), the clip's first hops, the tail of the loads
  auto generic_step = [&]() {
    const RawFrame<IN_MODE> cur = q0;
    q0 = q1;
    if (t + 2 < t_stop) load_raw(p, fbase + t + 2, lane, q1);
    v2f v[8];
    consume(cur, v);
    const long long i = t - ta;
    if (parks && i < 3 && t < tb) {
      // frame ta + i feeds hops ta + i .. ta + 2 of the wave before: slots 0 .. 2 - i, pieces {0,1,2}, {3,4}, {5}
      const int base = i == 0 ? 0 : (i == 1 ? 3 : 5);
#pragma unroll
      for (int s = 0; s < 3; ++s)
        if (s < 3 - i) park[(base + s) * 64] = make_float4(v[2 * s].x, v[2 * s].y, v[2 * s + 1].x, v[2 * s + 1].y);
    }
    if (t >= pe0 && t < pe1) emit_regs(t, acc[0], acc[1]);
    advance();
    ++t;
  };

  // steady state: own hop, full envelope, frame t + 2 exists and is this wave's to load, nothing to park
  long long fast_begin = ta + 3 > pe0 ? ta + 3 : pe0;
  if (fast_begin < 3) fast_begin = 3;
  long long fast_end = tb - 1;                                       // inclusive; the barrier sits at t == tb
  if (fast_end > t_stop - 3) fast_end = t_stop - 3;
  while (t < fast_begin && t < tb) generic_step();
  if (t <= fast_end) {
    auto fast_step = [&](RawFrame<IN_MODE>& q) {
      const RawFrame<IN_MODE> cur = q;
      load_raw(p, fbase + t + 2, lane, q);
      v2f v[8];
      consume(cur, v);
      float2* dst = reinterpret_cast<float2*>(yclip + (t - 2) * H);
      put_y(dst + lane, make_float2(acc[0].x * rcp[0].x, acc[0].y * rcp[0].y));
      put_y(dst + lane + 64, make_float2(acc[1].x * rcp[1].x, acc[1].y * rcp[1].y));
      advance();
      ++t;
    };
    while (t + 1 <= fast_end) {
      fast_step(q0);
      fast_step(q1);
    }
  }
  while (t < tb) generic_step();
  __syncthreads();                                                   // every wave's pieces are in LDS
  if (self_cool) {
    while (t < t_stop) generic_step();                               // the next tile's (or nobody's) first three frames
    // the clip's last hop has no frame of its own number
    if (t_stop == T && T >= pe0 && T < pe1 && t == T) emit_regs(T, acc[0], acc[1]);
  } else if (tb > ta) {
    // hops tb, tb + 1, tb + 2 (all inside the clip: the next wave holds at least three frames): the next wave's frames
    // one at a time, oldest first
    const float4* nx = park_all + (wave + 1) * 6 * 64 + lane;
    // piece (frame i, slot s) lands in hop i + s; its window samples are those of slot s
    auto add4 = [&](int hop, int s, const float4 pc) {
      acc[2 * hop] = __builtin_elementwise_fma((v2f){pc.x, pc.y}, ola_window(win, lane, 2 * s), acc[2 * hop]);
      acc[2 * hop + 1] = __builtin_elementwise_fma((v2f){pc.z, pc.w}, ola_window(win, lane, 2 * s + 1), acc[2 * hop + 1]);
    };
    add4(0, 0, nx[0 * 64]);
    add4(1, 1, nx[1 * 64]);
    add4(1, 0, nx[3 * 64]);
    add4(2, 2, nx[2 * 64]);
    add4(2, 1, nx[4 * 64]);
    add4(2, 0, nx[5 * 64]);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (tb + s >= pe0) emit_regs(tb + s, acc[2 * s], acc[2 * s + 1]);
  }
}

// K5: irFFT + window, frames out (no overlap-add): RealtimeSTFT/RealtimeDGT.invert
template <int IN_MODE>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void irfft1024_frames_kernel(InvParams p) {
  __shared__ float2 lds_all[WAVES_PER_BLOCK * kFftLdsFloat2PerWave + 512];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float2* lds = lds_all + wave * kFftLdsFloat2PerWave;
  float2* win = lds_all + WAVES_PER_BLOCK * kFftLdsFloat2PerWave;
  for (int i = threadIdx.x; i < 512; i += 64 * WAVES_PER_BLOCK) win[i] = scaled_window(p.window, i);
  __syncthreads();
  Twiddles tw;
  load_twiddles<true>(tw, p.tw, lane);
  const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
  long long f_end = f_begin + p.frames_per_block;
  if (f_end > p.total_frames) f_end = p.total_frames;
  for (long long f = f_begin + wave; f < f_end; f += WAVES_PER_BLOCK) {
    float2 v[8];
    float nyq;
    load_spectrum<IN_MODE>(p, f, lane, v, nyq);
    synth_frame(v, nyq, tw, win, lds, lane);
    float2* dst = reinterpret_cast<float2*>(p.y + f * N);
#pragma unroll
    for (int m = 0; m < 8; ++m) dst[lane + 64 * m] = v[m];
  }
}

}  // namespace at_hip

// ---------------------------------------------------------------------------
// host launchers (C++ linkage inside the library; the extern "C" ABI is capi.hip)
// ---------------------------------------------------------------------------
namespace at_hip {

// pass lengths of the reference's default bank at sr 44100 / n_fft 1024 in quads, one per nibble (band_bank.h).  The packed
// epilogue also hard-codes the feature row as 513 floats (its 1-KB block stream): the dispatch below asks for
// n_filters == 513 as well -- a 514..576-filter bank can have the same nine pass lengths
constexpr fqp_t kDefaultBankQuads = 0x001111223ull;
constexpr int kDefaultBankPasses = 9;      // 513 filters: seven passes of walks, two of empty filters
static bool bank_is(const BandBank* bank, fqp_t fqp, int n_passes) {
  if (bank->n_passes != n_passes) return false;
  for (int q = 0; q < bank->n_passes; ++q)
    if (bank->pass_len[q] != 4 * fqp_quads(fqp, q)) return false;
  return true;
}

// Counter pairs of the persistent kernels live behind the device's twiddle table (capi.hip: at_init allocates
// kTwiddleCount float2 + kTileCtrSlots pairs, zeroed).  Every launch takes the next pair of the ring; a kernel leaves
// its pair zeroed, so a pair is only ever in doubt if kTileCtrSlots launches are issued while one is still pending on
// another stream.
unsigned* tile_counter_slot(const float2* tw) {
  static std::atomic<unsigned> cursor{0};
  const unsigned slot = cursor.fetch_add(1, std::memory_order_relaxed) % kTileCtrSlots;
  return reinterpret_cast<unsigned*>(const_cast<float2*>(tw) + kTwiddleCount) + 2 * slot;
}

int launch_stft1024_fwd(const float* x, long long B, long long L, long long clip_stride, long long T, int hop,
                        int center, const float* window, const float2* tw, float2* out, float* phase,
                        hipStream_t stream) {
  FwdParams p;
  p.x = x; p.window = window; p.tw = tw; p.out = out; p.phase = phase;
  p.B = B; p.L = L; p.clip_stride = clip_stride; p.T = T; p.hop = hop; p.center = center;
  p.total_frames = B * T;
  if (p.total_frames == 0) return 0;
  // persistent-ish grid: up to 4 blocks (16 waves) per CU, each block a contiguous run of frames
  long long max_blocks = (long long)num_cus() * 4;
  long long fpb = (p.total_frames + max_blocks - 1) / max_blocks;
  fpb = ((fpb + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK) * WAVES_PER_BLOCK;
  if (fpb < 2 * WAVES_PER_BLOCK) fpb = 2 * WAVES_PER_BLOCK;
  p.frames_per_block = fpb;
  long long blocks = (p.total_frames + fpb - 1) / fpb;
  if (phase)
    hipLaunchKernelGGL(stft1024_fwd_kernel<true>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  else
    hipLaunchKernelGGL(stft1024_fwd_kernel<false>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_stft1024_h256_fwd(const float* x, long long B, long long L, long long clip_stride, long long T,
                             const float* window, const float2* tw, float2* out, float* phase, const BandBank* bank,
                             float* feat, const float* offset, const float* scale, float eps, int contrast, int power2,
                             int feat_channel_major, hipStream_t stream, const PolarOut* polar, int hop) {
  if (hop != 256 && hop != 128 && hop != 512) return -1;
  // at hop 128 / 512 the fused epilogue exists in its two plain forms (spectrum + features, features only); the
  // channel-major (MFCC) and Polar variants are built for the reference's hop 256
  if (hop != 256 && bank && (polar || feat_channel_major)) return -1;
  FwdRunParams p = {};
  p.x = x; p.window = window; p.tw = tw; p.out = out; p.phase = phase;
  p.B = B; p.L = L; p.clip_stride = clip_stride; p.T = T;
  if (bank) {
    p.bank = *bank; p.feat = feat; p.offset = offset; p.scale = scale; p.eps = eps; p.contrast = contrast;
    p.power2 = power2; p.feat_channel_major = feat_channel_major;
  }
  if (polar) {
    if (!bank || out) return -1;
    p.phase = polar->phase; p.feat_ld = polar->feat_ld; p.phase_ld = polar->phase_ld;
    p.ph_offset = polar->ph_offset; p.ph_scale = polar->ph_scale;
  }
  if (B * T == 0) return 0;
  size_t dyn_lds = 0;
  if (bank) {
    for (int q = 0; q < bank->n_passes; ++q) dyn_lds += (size_t)64 * bank->pass_len[q] * sizeof(float);
    if (dyn_lds > kMaxBandFloats * sizeof(float)) return -2;
    dyn_lds += (size_t)2 * 64 * bank->n_passes * sizeof(int);   // lane_start, lane_filter
  }
  // plain forward: 4 waves per block, twiddles in registers (3 waves per SIMD).  Fused: 8 waves share the band
  // table and read their twiddles from a workgroup LDS copy, which frees 44 VGPRs for a 4th wave per SIMD to
  // cover the epilogue's LDS round trips (4 % faster than the 3-wave form, A/B on one device).
  int NW = 4;
  bool default_bank_fixed = false;
  bool fq_logpow = false;
  bool persistent = false;
  void (*kernel)(FwdRunParams) = nullptr;
  if (!bank) {
    if (hop == 128)
      kernel = phase ? stft1024_h256_fwd_kernel<true, 0, 4, false, 0, false, 1> : stft1024_h256_fwd_kernel<false, 0, 4, false, 0, false, 1>;
    else if (hop == 512)
      kernel = phase ? stft1024_h256_fwd_kernel<true, 0, 4, false, 0, false, 4> : stft1024_h256_fwd_kernel<false, 0, 4, false, 0, false, 4>;
    else
      kernel = phase ? stft1024_h256_fwd_kernel<true, 0, 4, false> : stft1024_h256_fwd_kernel<false, 0, 4, false>;
  } else {
    NW = 8;
    if (!out && !polar && feat_channel_major && bank->n_passes == 1) kernel = stft1024_h256_fwd_kernel<false, 2, 8, true, 1>;
    else if (!out && !polar && feat_channel_major && bank->n_passes == 2) kernel = stft1024_h256_fwd_kernel<false, 2, 8, true, 2>;
    else if (!out && polar) kernel = stft1024_h256_fwd_kernel<false, 2, 8, true, 0, true>;
    else if (!out) kernel = stft1024_h256_fwd_kernel<false, 2, 8, true>;
    else kernel = phase ? stft1024_h256_fwd_kernel<true, 1, 8, true> : stft1024_h256_fwd_kernel<false, 1, 8, true>;
    // row-major features of a one- / two-pass bank at the default hop: passes unrolled, lane constants hoisted
    if (hop == 256 && !polar && !phase && !feat_channel_major && bank->n_passes <= 2) {
      if (bank->n_passes == 1)
        kernel = out ? stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 1> : stft1024_h256_fwd_kernel<false, 2, 8, true, 0, false, 2, 1>;
      else
        kernel = out ? stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 2> : stft1024_h256_fwd_kernel<false, 2, 8, true, 0, false, 2, 2>;
    }
    // ... and with the walk lengths of the headline bank (128 mel filters at 44.1 kHz: 8 + 2 quads), log1p and |X|:
    // the fixed-length epilogue
    if (hop == 256 && !polar && !phase && !feat_channel_major && bank->n_passes == 2 && contrast == 1 && !power2 &&
        bank->pass_len[0] == 32 && bank->pass_len[1] == 8 && variant(kVarEpilogue) == 0)
      kernel = out ? stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 2, 8, 2>
                   : stft1024_h256_fwd_kernel<false, 2, 8, true, 0, false, 2, 2, 8, 2>;
    // the log-mel of BASELINE configs[3] (log contrast, |X|^2, features only) on the same fixed-length epilogue
    if (hop == 256 && !polar && !phase && !feat_channel_major && !out && bank->n_passes == 2 && contrast == 2 && power2 &&
        bank->pass_len[0] == 32 && bank->pass_len[1] == 8 && variant(kVarEpilogue) == 0) {
      kernel = stft1024_h256_fwd_kernel<false, 2, 4, true, 0, false, 2, 2, 8, 2, false, false, 3, 2, true>;
      fq_logpow = true;
    }
    // MelSpectrogram (the reference's MFCC: |X|^2 on the 128-filter bank, no contrast, channel-major (.., N, T) output,
    // spectrum never stored) on the same fixed-length epilogue, its two results parked in the register windows
    if (hop == 256 && !polar && !phase && feat_channel_major && !out && bank->n_passes == 2 && contrast == 0 && power2 &&
        bank->pass_len[0] == 32 && bank->pass_len[1] == 8 && variant(kVarEpilogue) == 0) {
      // 4-wave blocks at three waves per SIMD (142 registers); with the pass twiddles in registers as well (HYB = 3) the
      // two windows no longer fit and spill (0.75 ms against 0.72; the run-time-length epilogue: 0.755)
      kernel = stft1024_h256_fwd_kernel<false, 2, 4, true, 2, false, 2, 2, 8, 2, false, false, 0, 0, true>;
      fq_logpow = true;
    }
    // The reference's default bank -- Magnitude() at sr 44100: 404 non-empty filters of 513 in seven passes of 3, 2, 2, 1,
    // 1, 1, 1 quads (and two of empty filters) -- with log1p and |X|, FEATURES ONLY (the README chain's forward): the
    // packed fixed-length epilogue, 0.92 -> 0.78-0.84 ms per 1024 clips.  The spectrum-storing form keeps the generic
    // epilogue: it is bound by the memory system's rate for one read and two write streams (~4.1 TB/s), and neither fewer
    // instructions (-31 %) nor fewer write requests (115 -> 96 per frame) nor aligned spectrum blocks moved it
    // (1.25-1.34 ms in every combination, same boxes: profiles/r04_default_bank_513.md).
    if (hop == 256 && !polar && !phase && !feat_channel_major && !out && contrast == 1 && !power2 &&
        bank->n_filters == F && bank_is(bank, kDefaultBankQuads, kDefaultBankPasses) && (((uintptr_t)feat) & 15) == 0 &&
        variant(kVarEpilogue) == 0) {
      kernel = stft1024_h256_fwd_kernel<false, 2, 8, true, 0, false, 2, 0, 0, 0, false, false, 0, 1, false, false, kDefaultBankQuads, kDefaultBankPasses>;
      default_bank_fixed = true;
    }
    if (hop == 128) {
      if (!out) kernel = stft1024_h256_fwd_kernel<false, 2, 8, true, 0, false, 1>;
      else kernel = phase ? stft1024_h256_fwd_kernel<true, 1, 8, true, 0, false, 1> : stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 1>;
    } else if (hop == 512) {
      if (!out) kernel = stft1024_h256_fwd_kernel<false, 2, 8, true, 0, false, 4>;
      else kernel = phase ? stft1024_h256_fwd_kernel<true, 1, 8, true, 0, false, 4> : stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 4>;
    }
  }
  // Runs of at least 8 frames.  The planner minimises rounds x (run length + per-run overhead), counting 1024 / hop - 3
  // >= 1 frames of overhead per run start: a full batch still gets long runs (1024 clips: 173 frames each, one round),
  // while a handful of clips is cut into many short runs that fill the idle chip -- one clip 63 -> 29 us, eight clips
  // 66 -> 31 us (it used to stop at 24-frame runs, i.e. seven waves per second of audio).
  // Aligned stream stores (template flags AL / NT) for the two headline forms at the default hop: the plain forward and
  // the fixed-length fused epilogue.  ACIDS_FWD_STORES = rows | aligned | aligned_nt picks the form for A/B runs.
  {
    static const int store_mode = [] {
      const char* e = dev_env("ACIDS_FWD_STORES");        // dev builds only
      if (!e) return 2;
      return !strcmp(e, "rows") ? 0 : !strcmp(e, "aligned") ? 1 : 2;
    }();
    const bool al_ok = store_mode != 0 && hop == 256 && out && !phase && !polar && (((uintptr_t)out) & 511) == 0;
    // persistent workgroups (PW): measured slower than one long run per wave on the product kernels
    // (profiles/r04_launch_shape.md); a development variant
    static const bool pw_mode = [] { const char* e = dev_env("ACIDS_FWD_PW"); return e && e[0] == '1'; }();
    if (al_ok && !bank) {
      NW = 8;
      kernel = store_mode == 1 ? stft1024_h256_fwd_kernel<false, 0, 8, true, 0, false, 2, 0, 0, 0, true, false>
                               : stft1024_h256_fwd_kernel<false, 0, 8, true, 0, false, 2, 0, 0, 0, true, true>;
      if (store_mode == 2 && pw_mode) {
        kernel = stft1024_h256_fwd_kernel<false, 0, 8, true, 0, false, 2, 0, 0, 0, true, true, 0, 1, false, true>;
        persistent = true;
      }
    } else if (al_ok && kernel == (void (*)(FwdRunParams))stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 2, 8, 2>) {
      kernel = store_mode == 1 ? stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 2, 8, 2, true, false>
                               : stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 2, 8, 2, true, true>;
      if (store_mode == 2 && pw_mode) {
        kernel = stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 2, 8, 2, true, true, 0, 1, false, true>;
        persistent = true;
      }
#ifdef AT_DEV_SWITCHES
      // dev A/B (round 5, energy): pass twiddles (1) / + window (3) in registers, three waves per SIMD in 4-wave blocks
      if (const char* e = dev_env("ACIDS_FWD_HYB")) {
        if (atoi(e) == 1) { kernel = stft1024_h256_fwd_kernel<false, 1, 4, true, 0, false, 2, 2, 8, 2, true, true, 1>; NW = 4; }
        if (atoi(e) == 3) { kernel = stft1024_h256_fwd_kernel<false, 1, 4, true, 0, false, 2, 2, 8, 2, true, true, 3>; NW = 4; }
      }
#endif
    }
    // Features only (the spectrum never stored) is bound by the LDS and by instruction issue, not by HBM: with both
    // pass-twiddle tables and the window in registers (HYB = 3: 30 fewer LDS reads per frame) at three waves per SIMD --
    // three 4-wave blocks per CU -- it runs 4 % faster than with four waves that read everything from LDS (0.642 ->
    // 0.615 ms, same box, alternating runs).  The spectrum-storing forms did not move with any HYB setting (fused 0.867
    // / 0.866 / 0.869 ms, plain 0.756 / 0.758 / 0.779): their waves wait on store issue, and the plain one loses its
    // fifth and sixth wave.  5 waves per SIMD (10-wave blocks, 96 registers, 2 spilled): 0.65 -> 0.69 ms.
    if (kernel == (void (*)(FwdRunParams))stft1024_h256_fwd_kernel<false, 2, 8, true, 0, false, 2, 2, 8, 2> &&
        !dev_env("ACIDS_FWD_NOHYB")) {
      NW = 4;
      kernel = stft1024_h256_fwd_kernel<false, 2, 4, true, 0, false, 2, 2, 8, 2, false, false, 3>;
    }
    if (fq_logpow) NW = 4;
  }
  if (default_bank_fixed) dyn_lds += (size_t)64 * ((kDefaultBankPasses + 3) / 4 * 4) * sizeof(int);   // packed descriptors
  const long long slots = resident_waves(kernel, 64 * NW, dyn_lds);
  const long long fpr = plan_units_per_run(B, T, slots, 8, hop == 128 ? 5 : 1);
  p.frames_per_run = fpr;
  // The plain forward (no epilogue) with work for every slot several times over: 16-frame runs.  One workgroup's eight
  // runs are then a 128-frame tile of the output stream and the hardware dispatches the tiles in address order -- the
  // compact, advancing write front of profiles/r04_launch_shape.md: 0.75 -> 0.72 ms per 1024 clips, measured again under
  // the wave priorities (same-box, r04y).  The fused kernels lose at every run length below the planner's (their run
  // start costs 2.4-3.4 us of wave time) and keep one long run per wave.
  if (!bank && hop == 256) p.frames_per_run = short_runs_if_full(B, T, slots, fpr, 16);
  if (const char* e = dev_env("ACIDS_FWD_FPR")) {     // dev builds: run length A/B, clamped to what the kernels assume
    const long long v = atoll(e);
    if (v >= 8 && v <= T) p.frames_per_run = v;
  }
  p.runs_per_clip = (T + p.frames_per_run - 1) / p.frames_per_run;
  const long long waves = B * p.runs_per_clip;
  if (dev_env("ACIDS_DEBUG_PLAN")) {
    hipFuncAttributes fa = {};
    (void)hipFuncGetAttributes(&fa, (const void*)kernel);
    fprintf(stderr, "[plan fwd] NW %d slots %lld (regs %d, static lds %zu, dyn %zu) frames/run %lld runs/clip %lld waves %lld\n", NW,
            slots, fa.numRegs, fa.sharedSizeBytes, dyn_lds, p.frames_per_run, p.runs_per_clip, waves);
  }
  long long blocks = (waves + NW - 1) / NW;
  if (persistent) {
    // short runs (the planner's figure is one long run per resident wave), tiles of NW runs, as many workgroups as fit
    long long g = 8;
    if (const char* e = dev_env("ACIDS_FWD_PW_RUN")) g = atoll(e) >= 2 ? atoll(e) : 8;
    if (g > T) g = T;
    p.frames_per_run = g;
    p.runs_per_clip = (T + g - 1) / g;
    const long long tiles = (B * p.runs_per_clip + NW - 1) / NW;
    if (tiles >= (1LL << 31)) return -1;
    p.n_tiles = (unsigned)tiles;
    p.tile_ctr = tile_counter_slot(tw);
    blocks = slots / NW;
    if (blocks > tiles) blocks = tiles;
  }
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * NW), dyn_lds, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_istft1024_ola(const float2* X, const float* mag, const float* phase, long long B, long long T, int hop,
                         const float* window, const float* env16, const float2* tw, float* y, hipStream_t stream,
                         const float2* gl_tprev, float gl_mom) {
  InvParams p = {};
  p.X = X; p.mag = mag; p.phase = phase; p.window = window; p.env = env16; p.tw = tw; p.y = y;
  p.tprev = gl_tprev; p.gl_mom = gl_mom;
  p.B = B; p.T = T;
  const long long nslots = T - 1;
  if (B == 0 || nslots <= 0) return 0;
  // two frames in flight per wave, twiddles in registers: 2 waves per SIMD.  (One frame in flight at 3 waves,
  // or LDS twiddles at 3-4 waves, all land within 3 % of each other: the kernel sits on its memory floor.)
  void (*kernel)(InvParams) = nullptr;
  if (hop == 128) kernel = X ? istft1024_ola_kernel<IN_COMPLEX, 2, false, 1> : istft1024_ola_kernel<IN_POLAR, 2, false, 1>;
  else if (hop == 256) kernel = X ? istft1024_ola_kernel<IN_COMPLEX, 2, false, 2> : istft1024_ola_kernel<IN_POLAR, 2, false, 2>;
  else if (hop == 512) kernel = X ? istft1024_ola_kernel<IN_COMPLEX, 2, false, 4> : istft1024_ola_kernel<IN_POLAR, 2, false, 4>;
  else return -1;
  if (gl_tprev) {   // Griffin-Lim update at load time: X = rebuilt, mag = target magnitudes; one frame in flight (40 VGPRs each)
    if (!X || !mag) return -1;
    kernel = hop == 128 ? istft1024_ola_kernel<IN_GL, 1, true, 1>
           : hop == 256 ? istft1024_ola_kernel<IN_GL, 1, true, 2> : istft1024_ola_kernel<IN_GL, 1, true, 4>;
  }
  // Full batches at the default hop: tiles of consecutive frames per workgroup, dispatched in address order, the overlap
  // state handed from wave to wave through LDS (istft1024_tile_kernel).  "Full" = at least two tiles for every workgroup
  // the chip holds; anything smaller keeps the planner's long runs below.  Same bits either way.
  if (hop == 256 && !gl_tprev && T >= 64 && variant(kVarIstftRuns) == 0) {
    int nw = 4, occ = 2;
    long long target = 175;                                  // frames per tile aimed at (same-box A/B of 125 ... 350: gpurun_out r05d)
    if (const char* e = dev_env("ACIDS_ISTFT_TILE")) {      // dev builds: "<waves>:<frames per tile>[:<waves per SIMD>]", "0" = long runs
      nw = atoi(e);
      if (const char* c = strchr(e, ':')) {
        target = atoll(c + 1);
        if (const char* c2 = strchr(c + 1, ':')) occ = atoi(c2 + 1);
      }
    }
    if (nw == 4 || nw == 8) {
      void (*tk)(InvParams) = nw == 8 ? (X ? istft1024_tile_kernel<IN_COMPLEX, 8, 2> : istft1024_tile_kernel<IN_POLAR, 8, 2>)
                                      : (X ? istft1024_tile_kernel<IN_COMPLEX, 4, 2> : istft1024_tile_kernel<IN_POLAR, 4, 2>);
      if (nw == 4 && occ == 3) tk = X ? istft1024_tile_kernel<IN_COMPLEX, 4, 3> : istft1024_tile_kernel<IN_POLAR, 4, 3>;
      if (target < 8 * nw) target = 8 * nw;
      const long long tiles_per_clip = (T + target - 1) / target;
      const long long want = (T + tiles_per_clip - 1) / tiles_per_clip;      // frames per tile, balanced over the clip
      long long n = (want + 3 + nw - 1) / nw;                                // frames per wave; the last wave n - 3
      if (n < 6) n = 6;
      const long long tile_frames = nw * n - 3;
      const long long tpc = (T + tile_frames - 1) / tile_frames;
      const long long resident = resident_waves(tk, 64 * nw, 0) / nw;
      if (B * tpc >= 2 * resident && B * tpc < (1LL << 31)) {
        p.slots_per_run = n;
        p.frames_per_block = tile_frames;
        p.runs_per_clip = tpc;
        hipLaunchKernelGGL(tk, dim3((unsigned)(B * tpc)), dim3(64 * nw), 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -5;
      }
    }
  }
  // runs of >= 8 hop slots (a run synthesises n_fft/hop - 1 frames more than it emits slots: the planner's overhead term)
  const long long slots = resident_waves(kernel, 64 * WAVES_PER_BLOCK, 0);
  const long long spr = plan_units_per_run(B, nslots, slots, 8, 1024 / hop - 1);
  p.slots_per_run = spr;
  if (const char* e = dev_env("ACIDS_ISTFT_SPR")) {   // dev builds: run length A/B
    const long long v = atoll(e);
    if (v >= 8 && v <= nslots) p.slots_per_run = v;
  }
  p.runs_per_clip = (nslots + p.slots_per_run - 1) / p.slots_per_run;
  const long long waves = B * p.runs_per_clip;
  const long long blocks = (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int launch_irfft1024_frames(const float2* X, const float* mag, const float* phase, long long nframes,
                            const float* window, const float2* tw, float* y, hipStream_t stream) {
  InvParams p = {};
  p.X = X; p.mag = mag; p.phase = phase; p.window = window; p.tw = tw; p.y = y;
  p.total_frames = nframes;
  if (nframes == 0) return 0;
  long long max_blocks = (long long)num_cus() * 4;
  long long fpb = (nframes + max_blocks - 1) / max_blocks;
  fpb = ((fpb + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK) * WAVES_PER_BLOCK;
  p.frames_per_block = fpb;
  long long blocks = (nframes + fpb - 1) / fpb;
  if (X)
    hipLaunchKernelGGL(irfft1024_frames_kernel<IN_COMPLEX>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  else
    hipLaunchKernelGGL(irfft1024_frames_kernel<IN_POLAR>, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace at_hip
