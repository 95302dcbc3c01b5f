/*
 * acids_hip.h -- C ABI of libacids_hip.so, the MI355X (gfx950) implementation
 * of the spectral hot path of acids_transforms.
 *
 * The reference is pure Python on torch CPU ops and has no FFI layer
 * (SURVEY.md 8b); every entry point below therefore cites the reference
 * *call site* it replaces (paths relative to acids_transforms/).  A maintainer
 * binds these with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - complex64 arrays are interleaved float pairs (re, im);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     all work is asynchronous on that stream; nothing allocates or
 *     synchronises except at_init();
 *   - return value: 0 on success, a negative AT_E* code otherwise; nothing
 *     throws across the boundary;
 *   - shapes use the reference's conventions: audio (B, L) float32, spectra
 *     (B, T, F) with F = n_fft/2 + 1, T = 1 + L / hop (center=True).
 */
#ifndef ACIDS_HIP_H
#define ACIDS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AT_OK 0
#define AT_EINVAL (-1)     /* bad argument (size, null pointer, alignment) */
#define AT_EUNSUPPORTED (-2) /* valid in the reference, not implemented here (e.g. non power-of-two n_fft) */
#define AT_ENOTINIT (-3)   /* at_init() was not called for the current device */
#define AT_EWORKSPACE (-4) /* workspace too small */
#define AT_ELAUNCH (-5)    /* HIP launch / runtime error */

/* Returns 4.  History: 4 in round 4 (at_set_variant / at_get_variant).  Earlier history: 2 early in round 2 (at_sinebank_realtime gained the window argument; at_mel_*bf16*, at_oadd_push
 * added); 3 later in round 2 (at_pghi_realtime_seeded, at_phase_*_strided / _polar, at_cartesian_*).  Round 3 changed
 * kernels only: no signature moved. */
int at_abi_version(void);
const char *at_error_string(int code);

/* Kernel variants.  The library decides from a call's arguments which kernel runs; it reads NO environment variables
 * (the A/B switches of tools/ab*.sh exist only in -DAT_DEV_SWITCHES builds).  Where two kernels compute the same
 * result -- a form specialised for the headline shapes and the generic one -- this process-wide table lets a caller force
 * the generic one; the parity tests use it to compare the two.  No reference counterpart (the reference has one
 * implementation of everything).  value: 0 = default; returns AT_EINVAL for an unknown variant or value. */
#define AT_VARIANT_EPILOGUE 0          /* 1: generic mel epilogue / projection even for the 128-mel headline bank */
#define AT_VARIANT_FRAME_KERNELS 1     /* 1: frame-at-a-time forward at n_fft 512 / 2048 / 4096 (no sliding window) */
#define AT_VARIANT_SMALL_PROJECTION 2  /* 1: row kernel instead of the matrix-core form of the K <= 128 projection */
#define AT_VARIANT_SCAN_LAYOUT 3       /* 1: flattened columns instead of one block per clip in the phase scans */
#define AT_VARIANT_PGHI_KERNEL 4       /* 1: winner-bit offline heap kernel; 2: single-lane heap kernels; 3: realtime
                                        * flood on the heap only; 4: realtime flood on the rank bitmap / heap, without the
                                        * wavefront-parallel scan path (round 5) that is tried first by default */
#define AT_VARIANT_ISTFT_RUNS 5        /* 1: n_fft-1024 inverse as one long run per wave even for full batches (no workgroup
                                        * tiles with the overlap state handed over through LDS) */
int at_set_variant(int which, int value);
int at_get_variant(int which);

/* One-time per-device setup (twiddle tables).  Synchronous; call before capture. */
int at_init(int device);

/* ---- K1/K2/K4: forward ------------------------------------------------ */
/* torch.stft(x, n_fft, hop, window, center=True, pad_mode="reflect",
 * onesided=True, return_complex=True).transpose(-2,-1)
 *     replaces transforms/stft.py:98-104 and transforms/dgt.py:64-70.
 * center=0: frame t starts at sample t*hop of its clip, zero padded past L
 *     (utils/misc.py:148-165 frame()) then rfft(x*window):
 *     replaces stft.py:249-253 / dgt.py:285-289 on OverlapAdd.forward output
 *     (oadd.py:69-74) without materialising the frames.
 * phase (optional, may be NULL): atan2(im, re) of every bin = the reference's
 *     phase_buffer side effect (stft.py:103).
 * Speed, not correctness (round 3): at n_fft 512 / 1024 / 2048 / 4096 with hop = n_fft / 4, center = 1, no phase
 *     output, clips starting on 16-byte boundaries (8 for n_fft <= 1024) and `out_complex` 512-byte aligned, the
 *     sliding-window kernels with aligned non-temporal stream stores run (0.59 - 0.64 of the HBM roofline); any other
 *     arguments take the frame-at-a-time / row-store kernels with identical results to 1e-6. */
int at_stft_forward(const float *x, int64_t B, int64_t L, int64_t clip_stride, int64_t T, int n_fft, int hop,
                    int center, const float *window, float *out_complex, float *phase, void *stream);

/* Fused forward for Compose(STFT|DGT -> Magnitude(mel)) and for MFCC (n_fft = 1024, hop = 256):
 * the same framing + rFFT kernel additionally emits normalise(contrast(|X|^p @ bank)) from registers, so
 * the spectrum is not re-read (spectral_repr.py:215-226 / mel.py:43-44,68-73 behind stft.py:98-104).
 * The bank is passed in banded form, as the walk of the kernel's epilogue: in pass q (n_passes <= 16) lane l
 * of a wavefront sums filter lane_filter[q*64 + l] (-1: none) over pass_len_host[q] bins starting at bin
 * lane_start[q*64 + l]; both arrays are DEVICE int32[n_passes*64], pass_len_host is a HOST array of
 * multiples of 4 (<= 128), lane_start entries are multiples of 4 with lane_start + pass_len <= 640.
 * band_weights (DEVICE, 16-byte aligned): the 4 weights lane l applies in step j of pass q are the floats at
 * ((quad_base[q] + j)*64 + l)*4, quad_base[q] = sum_{q' < q} pass_len[q']/4; zero outside the filter's band;
 * 64 * sum(pass_len) <= 8192 floats.  (utils/banded.py builds these from a dense bank and chooses the lanes so
 * that the LDS reads of the walk are bank-conflict free.)
 * Exact for any bank whose columns are zero outside their band; dense banks use at_mel_project.
 * out_complex_or_null == NULL: features only (MFCC).
 * feat: (B*T, n_filters), or (B, n_filters, T) when feat_channel_major.
 * n_fft = 1024; hop = 256, or 128 / 512 with row-major features (else AT_EUNSUPPORTED).
 * n_fft = 2048 (window 16-byte aligned) or 512, any hop (here lane_start + pass_len <= n_fft/2 + 1 + 128): features
 * only -- out_complex_or_null and phase_or_null must be NULL -- either layout: MelSpectrogram / MFCC in one kernel
 * (stft2048.hip, stft512.hip), the spectrum is never written. */
int at_stft_mel_forward(const float *x, int64_t B, int64_t L, int64_t clip_stride, int64_t T, int n_fft, int hop,
                        const float *window, const int32_t *lane_filter, const int32_t *lane_start,
                        const float *band_weights, int n_filters, int n_passes, const int32_t *pass_len_host,
                        int contrast, int power2, const float *offset, const float *scale, float eps,
                        float *out_complex_or_null, float *phase_or_null, float *feat, int feat_channel_major,
                        void *stream);

/* Compose(STFT -> Polar) in one kernel (spectral_repr.py:511-523 behind stft.py:98-104): framing + rFFT, then
 * out[r, 0, :] = normalise(contrast(|X| @ bank)) (banded bank with F = 513 filters, tables as above) and
 * out[r, 1, :] = normalise(angle X), r = clip * T + frame; out_stacked: (B*T, 2, F).  The complex spectrum
 * is never written. */
int at_stft_polar_forward(const float *x, int64_t B, int64_t L, int64_t clip_stride, int64_t T, int n_fft, int hop,
                          const float *window, const int32_t *lane_filter, const int32_t *lane_start,
                          const float *band_weights, int n_filters, int n_passes, const int32_t *pass_len_host,
                          int contrast, const float *mag_offset, const float *mag_scale, float eps,
                          const float *phase_offset, const float *phase_scale, float *out_stacked, void *stream);

/* ---- K3/K5/K15: inverse ------------------------------------------------ */
/* 2^R x hop table of window^2 sums (R = n_fft / hop = 2, 4 or 8: the frames that overlap one hop) used by the fused
 * at_istft path, n_fft = 1024 with hop = 512, 256 (the reference's default) or 128 -- torch.istft's window envelope
 * (stft.py:126-127) for every combination of present / missing neighbour frames.  env16: 2^R * hop floats.
 * Other ratios: AT_EUNSUPPORTED (at_istft then takes its workspace path). */
int at_istft_envelope_table(const float *inv_window, int n_fft, int hop, float *env16, void *stream);

size_t at_istft_workspace_bytes(int64_t B, int64_t T, int n_fft, int hop);

/* torch.istft(X.transpose(-2,-1), n_fft, hop, window=inv_window, onesided=True)
 *     replaces stft.py:120-128, dgt.py:86-93 (complex input: X != NULL), and
 *     `x * exp(1j*phase)` + istft, stft.py:157-161 / dgt.py:152-154
 *     (polar input: X == NULL, mag and phase given).
 * y: (B, hop*(T-1) + (n_fft & 1)) -- torch.istft trims n_fft/2 (floor) at both ends.  Any n_fft in [2, 16384] (odd
 * sizes below 8192): powers of two on the kernels named below, everything else on the mixed-radix kernels of
 * stft_mixed.hip (the same holds for at_stft_forward and at_irfft_frames).  env16 (at_istft_envelope_table) is
 * required for n_fft = 1024, 512, 2048 and 4096 with hop = n_fft/8, n_fft/4 or n_fft/2 -- the fused kernels, for which
 * at_istft_workspace_bytes is 0 -- and may be NULL otherwise.  n_fft = 128, 256, 512, 2048 and 4096 run on the
 * register FFT core (stft_small.hip, stft512.hip, stft2048.hip, stft4096.hip), the other powers of two on the generic
 * LDS kernel.  n_fft = 1024, hop = 256: a batch with at least two tiles of ~175 frames for every workgroup the chip
 * holds runs as workgroup tiles (the overlap state crosses the cuts between a workgroup's waves through LDS), anything
 * smaller as one run per wave; the window is taken inside the accumulation (fma, frame order) in both, so a clip's bits
 * do not depend on the batch it rides in (AT_VARIANT_ISTFT_RUNS = 1 forces the long runs). */
int at_istft(const float *X_complex, const float *mag, const float *phase, int64_t B, int64_t T, int n_fft, int hop,
             const float *inv_window, const float *env16, float *y, void *workspace, size_t workspace_bytes,
             void *stream);

/* torch.fft.irfft(X) * inv_window on (nframes, F) -> (nframes, n_fft)
 *     replaces stft.py:260-266,308-310 / dgt.py:296-302,325-328. */
int at_irfft_frames(const float *X_complex, const float *mag, const float *phase, int64_t nframes, int n_fft,
                    const float *inv_window, float *frames, void *stream);

/* ---- K8-K12: magnitude / mel projection ---------------------------------- */
/* a_kind: 0 = complex64 input, |.| taken on load; 1 = complex64, |.|^2 (power=2);
 *         2 = float32 input; 3 = float32 input, |.| taken.   contrast: 0 none, 1 log1p, 2 log, 3 log10.
 * forward (inverse=0):  out = normalise(contrast(A' @ bank))
 *     replaces  x.abs(); torch.matmul(mag, mel_bank); contrast; Normalize.forward
 *     (spectral_repr.py:215-226, norm.py:40-41) and, with a_kind=1 and
 *     T_transposed=T, torchaudio MelSpectrogram's |stft|^2 @ fbank with its
 *     channel-major (..., n_mels, T) output (mel.py:43-44, 68-73).
 * inverse (inverse=1, a_kind=2): out = invert_contrast(A*scale+offset) @ bank
 *     replaces  Normalize.invert; invert_contrast; matmul(mag, inverse_mel_bank)
 *     (spectral_repr.py:228-240, norm.py:43-44).
 * A: rows x K with row stride lda (elements); bank: K x N row-major, stride ldb;
 * out: rows x N with stride ld_out, or (rows/T, N, T) when T_transposed > 0.
 * offset/scale: device scalars, both NULL = no normalisation.  Exact fp32 MFMA. */
int at_mel_project(const void *A, int a_kind, int64_t rows, int64_t lda, int K, const float *bank, int ldb, int N,
                   int contrast, int inverse, const float *offset, const float *scale, float eps, float *out,
                   int64_t ld_out, int64_t T_transposed, void *stream);

/* The forward projection as a dense bf16 MFMA GEMM with fp32 accumulation (v_mfma_f32_32x32x16_bf16): BASELINE
 * config 5's "bf16 MFMA mel".  Same chain as at_mel_project with inverse = 0 -- normalise(contrast(A' @ bank)),
 * spectral_repr.py:215-226 -- but A' (|x|, |x|^2 or the real input) and the bank are rounded to bf16 (nearest even)
 * before the products; sums are fp32.  ~4e-3 relative to the fp32 chain: opt-in only (Magnitude(bank_dtype="bf16")),
 * never the default.  The bank is passed as the packed operand image written by at_mel_bf16_pack_bank
 * (at_mel_bf16_bank_bytes(K, N) bytes, 16-byte aligned): [ceil32(N)][ceil16(K)] bf16, k contiguous, zero padded.
 * out: rows x N fp32, row stride ld_out.  K up to 4096 (one 32-column tile of the bank must fit in LDS), else
 * AT_EUNSUPPORTED. */
size_t at_mel_bf16_bank_bytes(int K, int N);
int at_mel_bf16_pack_bank(const float *bank, int K, int ldb, int N, void *bank_bf16, void *stream);
int at_mel_project_bf16(const void *A, int a_kind, int64_t rows, int64_t lda, int K, const void *bank_bf16, int N,
                        int contrast, const float *offset, const float *scale, float eps, float *out, int64_t ld_out,
                        void *stream);

/* Magnitude with mel=False: the same chains without the projection (n elements). */
int at_mag_pointwise(const void *A, int a_kind, int64_t n, int contrast, int inverse, const float *offset,
                     const float *scale, float eps, float *out, void *stream);

/* {min, max, sum, sum of squares} (fp64) of contrast(|A|) over n elements:
 *     Normalize.scale_data (norm.py:25-38) / Magnitude.scale_data (spectral_repr.py:242-245). */
size_t at_stats_workspace_bytes(void);
int at_stats(const void *A, int a_kind, int64_t n, int contrast, float eps, double *out4, void *workspace,
             size_t workspace_bytes, void *stream);

/* Normalize.forward (inverse=0): (x-offset)/scale;  Normalize.invert (inverse=1): x*scale+offset
 *     (norm.py:40-44); offset/scale are device scalars. */
int at_affine(const float *x, int64_t n, const float *offset, const float *scale, int inverse, float *out,
              void *stream);

/* ---- K13/K14: PGHI --------------------------------------------------------- */
/* DGT.modgabphasegrad (dgt.py:222-236) on B clamped-on-the-fly (T,F) magnitude
 * arrays: tgradw, fgradw (B,T,F).  spec_or_null receives clamp(mag, eps). */
int at_pghi_gradients(const float *mag, int64_t B, int T, int F, float gamma, int n_fft, int hop, float eps,
                      float *tgradw, float *fgradw, float *spec_or_null, void *stream);

size_t at_pghi_offline_workspace_bytes(int64_t B, int T, int F);

/* DGT.pghi + perform_hgi for every clip of a batch (dgt.py:137-141, 156-220) with
 * the binary-heap order of utils/heapq.py:9-59.  abstol is both the clamp floor
 * and the "visited" marker (the reference passes eps for both, dgt.py:157-162).
 * phase: (B,T,F).  npops_or_null: (B) pops per clip.  order_or_null: (B, T*F)
 * row*F+col of every pop in order (parity tests).
 * T * F <= 2^26 - 64 bins per clip (32-bit heap positions; 12 minutes of audio at the default sizes), else
 * AT_EUNSUPPORTED. */
int at_pghi_offline(const float *mag, int64_t B, int T, int F, float gamma, int n_fft, int hop, float tol,
                    float abstol, float *phase, void *workspace, size_t workspace_bytes, int64_t *npops_or_null,
                    int32_t *order_or_null, void *stream);

/* DGT.perform_hgi (dgt.py:168-220) on its own: the heap integration of B (T,F) magnitude arrays along gradients the
 * CALLER supplies (tgradw is applied along the bin axis, fgradw along the frame axis, as the reference's method uses
 * them), same pop order, same threshold rule (bins below tol * max are never visited).  mag is not modified (the
 * reference integrates in its argument); workspace as for at_pghi_offline. */
int at_pghi_integrate(const float *mag, const float *tgradw, const float *fgradw, int64_t B, int T, int F, float tol,
                      float abstol, float *phase, void *workspace, size_t workspace_bytes, int64_t *npops_or_null,
                      int32_t *order_or_null, void *stream);

size_t at_pghi_rt_workspace_bytes(int S, int n, int F);

/* RealtimeDGT.pghi (dgt.py:338-354, 378-466) for S streams: mag_hist (S,2,F),
 * mag (S,n,F), prev_phase (S,F), noise (S,n,F) standard-normal draws for the bins
 * at or below the tolerance (dgt.py:404-405) -> phase (S,n,F).  The time-border
 * rows the reference leaves uninitialised (dgt.py:388-394) are defined as 0.
 * tgradw/fgradw_or_null: optional (S,n+2,F) outputs. */
int at_pghi_realtime(const float *mag_hist, const float *mag, const float *prev_phase, const float *noise, int S,
                     int n, int F, float gamma, int n_fft, int hop, float tol, float eps, float *phase,
                     float *tgradw_or_null, float *fgradw_or_null, void *workspace, size_t workspace_bytes,
                     void *stream);

/* The same with the standard-normal draws of dgt.py:404-405 (torch.randn_like) made on the device, so that a captured
 * streaming step needs no generator launch: Philox-4x32-10 keyed by rng_state[0..1] (seed), counter = (bin index,
 * rng_state[2]); the call advances rng_state[2] by one.  rng_state: 4 x uint32 in device memory, owned by the caller
 * (one per session).  Statistical parity only: the draws are not torch's. */
int at_pghi_realtime_seeded(const float *mag_hist, const float *mag, const float *prev_phase, uint32_t *rng_state, int S,
                            int n, int F, float gamma, int n_fft, int hop, float tol, float eps, float *phase,
                            void *workspace, size_t workspace_bytes, void *stream);

/* RealtimeDGT.update_buffers (dgt.py:330-336) for x = mag*exp(i*phase):
 * hist_out = |x[-2:]| (or [hist_in[1], |x[-1]|] when n == 1), phase_out = angle(x[-1]).  hist_out may alias hist_in
 * (every bin reads its history before it writes it). */
int at_rt_update_buffers(const float *mag, const float *phase, int S, int n, int F, const float *hist_in,
                         float *hist_out, float *phase_out, void *stream);

/* ---- pointwise ----------------------------------------------------------- */
/* x.angle() on n complex64 values: the phase_buffer of stft.py:103 / dgt.py:69,
 * and hgi_phase_buffer of dgt.py:336. */
int at_angle(const float *x_complex, int64_t n, float *out, void *stream);

/* Cartesian.forward / invert (spectral_repr.py:403-428) in one pass: (rows, F) complex64 <-> (rows, 2, F) float32
 * with [r, 0, :] = (x.real - re_offset) / re_scale and [r, 1, :] = (x.imag - im_offset) / im_scale (Normalize,
 * norm.py:40-44; a NULL offset/scale pair: that half is not normalised); unpack applies y * scale + offset. */
int at_cartesian_pack(const float *x_complex, int64_t rows, int F, const float *re_offset, const float *re_scale,
                      const float *im_offset, const float *im_scale, float *stacked, void *stream);
int at_cartesian_unpack(const float *stacked, int64_t rows, int F, const float *re_offset, const float *re_scale,
                        const float *im_offset, const float *im_scale, float *out_complex, void *stream);

/* ---- Griffin-Lim building blocks (STFT's default inversion mode, stft.py:37,174-178) ---- */
/* One phase update of torchaudio.functional.griffinlim:  a = rebuilt - m*tprev (tprev may be NULL);
 * X = mag * a / (|a| + 1e-16), with m = momentum / (1 + momentum).  n complex elements. */
int at_griffinlim_update(const float *mag, const float *rebuilt_complex, const float *tprev_complex_or_null,
                         float momentum_over_1p, int64_t n, float *X_complex, void *stream);
/* The same update fused into the inverse: at_istft(mag * normalise(rebuilt - m' * tprev)) in one kernel, the updated
 * spectrum never written (one Griffin-Lim iteration = this + at_stft_forward).  n_fft = 1024, hop 128 / 256 / 512 with
 * env16 from at_istft_envelope_table; anything else: AT_EUNSUPPORTED (use at_griffinlim_update + at_istft). */
int at_istft_griffinlim(const float *mag, const float *rebuilt_complex, const float *tprev_complex_or_null,
                        float momentum_over_1p, int64_t B, int64_t T, int n_fft, int hop, const float *inv_window,
                        const float *env16, float *y, void *stream);
/* X = mag * z (z complex): the random initialisation of the same algorithm. */
int at_scale_complex(const float *mag, const float *z_complex, int64_t n, float *X_complex, void *stream);

/* ---- K6/K7: OverlapAdd streaming framer / overlap-add ---------------------- */
/* OverlapAdd.forward (oadd.py:69-74, 33-42): buf (S, buf_len) = [history | chunk | 0...],
 * hist_out = last `keep` samples of [history | chunk].  The caller takes the
 * (S, n, n_fft) frames as a strided view of buf (utils/misc.py:148-165).  C < keep (down to one hop per step) is
 * an extension: the reference's own state breaks there (oadd.py:41); the frame sequence is that of any other
 * chunking of the same sample stream. */
int at_oadd_forward(const float *x, const float *hist_in_or_null, int S, int64_t C, int keep, int64_t buf_len,
                    float *buf, float *hist_out, void *stream);
/* OverlapAdd.invert (oadd.py:90-104): frames (S, n, n_fft) + carried tail (S, keep)
 * -> out (S, (n-1)*hop + n_fft - keep) / gain, tail_out (S, keep) undivided.  tail_out may alias tail_in_or_null
 * (state updated in place); n = 1 emits one hop.  keep <= 16384. */
int at_oadd_invert(const float *frames, const float *tail_in_or_null, int S, int n, int n_fft, int hop, int keep,
                   const float *gain, float *out, float *tail_out, void *stream);
/* The input side of the same state kept in place (streaming sessions, hipGraph replay): buf (S, buf_len) holds
 * [history(keep) | chunk(C) | pad]; one call moves the last `keep` samples of [history | chunk] to the front and
 * writes the new chunk x (S, C) behind them (oadd.py:33-42, 69-74).  Any C >= 1; buf_len >= keep + C. */
int at_oadd_push(const float *x, int S, int64_t C, int keep, int64_t buf_len, float *buf, void *stream);

/* ---- K16: integer transforms -------------------------------------------------- */
/* torchaudio MuLawEncoding / MuLawDecoding as used by MuLaw (raw.py:282-283, 316):
 * codes int64 in [0, channels). */
int at_mulaw_encode(const float *x, int64_t n, int channels, int64_t *codes, void *stream);
int at_mulaw_decode(const int64_t *codes_i64, const float *codes_f32, int64_t n, int channels, float *x,
                    void *stream);
/* F.one_hot(x, classes) (misc.py:183-185, raw.py:288-291): out (n, classes) int64, or with
 * channel_major_inner = t > 0 the transposed (n/t, classes, t) layout of raw.py:289-290. */
int at_onehot(const int64_t *x, int64_t n, int classes, int64_t channel_major_inner, int64_t *out, void *stream);
/* Tensor.argmax(-1) (misc.py:188-189): first index of the row maximum. */
int at_argmax_last(const int64_t *x_i64, const float *x_f32, int64_t rows, int cols, int64_t *out, void *stream);

/* The same projection for a BANDED bank (mel banks, the reference's default 513 x 513 one and its inverse
 * included): the K inputs of a frame go through the prologue into LDS and every lane walks the band of
 * one filter per pass -- the walk tables are those of at_stft_mel_forward (utils/banded.py).  HBM-bound (input
 * row in, n_filters floats out) where the dense contraction is MFMA-bound.  Arguments as at_mel_project.
 * Limits (wider than the fused epilogue's): K <= 2112 (every n_fft up to 4096), n_passes <= 40, pass_len <= 512,
 * and weight table + lane tables + one LDS row per wave within 160 KB (AT_EUNSUPPORTED otherwise: use
 * at_mel_project).
 * phase_out != NULL (complex input): also writes normalise(angle(x)) with row stride ld_phase -- Polar.forward
 * (spectral_repr.py:432-439) fills its stacked (.., T, 2, F) result in one pass over the spectrum.
 * phase_in != NULL (inverse only): `out` is complex64, out[r, f] = acc * exp(i * (phase_in[r*ld_phase + f] *
 * phase_scale + phase_offset)) -- Polar.invert (spectral_repr.py:441-451) in one pass. */
int at_mel_project_banded(const void *A, int a_kind, int64_t rows, int64_t lda, int K, const int32_t *lane_filter,
                          const int32_t *lane_start, const float *band_weights, int n_filters, int n_passes,
                          const int32_t *pass_len_host, int contrast, int inverse, const float *offset,
                          const float *scale, float eps, float *out, int64_t ld_out, int64_t T_transposed,
                          float *phase_out, int64_t ld_phase, const float *phase_offset, const float *phase_scale,
                          const float *phase_in, void *stream);

/* normalise(x @ W) for a small real matrix (K <= 128, N <= 64): the DCT-II behind MFCC(n_mfcc).  K = 128 / 80 / 64 with
 * a 16-byte aligned x: fp32 MFMA tiles (v_mfma_f32_16x16x4_f32, fp32 accumulate); otherwise one wavefront per row with
 * W's columns in registers.  out: (rows, N), or channel-major (rows/T, N, T) when T_transposed > 0; x and out must not
 * overlap. */
int at_project_small(const float *x, int64_t rows, int K, const float *W, int N, const float *offset,
                     const float *scale, float *out, int64_t T_transposed, void *stream);

/* ---- phase-side representations ------------------------------------------ */
/* Scan along the frame axis of a (B, T, F) spectrum, one of X_complex / phase given (the angle is taken
 * inside), optional per-frame weight frame_window[T] and Normalize affine (device scalars) applied last:
 *   mode 0  unwrap(angle)                      utils/misc.py:12-26   (Phase(unwrap=True), spectral_repr.py:271-274)
 *   mode 1  IF "forward"   fdiff_forward(unwrap)/pi   rows [0, T-2]   utils/misc.py:65-68, spectral_repr.py:323-325
 *   mode 2  IF "backward"  fdiff_backward(unwrap)/-pi rows [1, T-1]   utils/misc.py:71-75, spectral_repr.py:320-322
 *   mode 3  IF "central"   fdiff_central(unwrap)/2pi  rows [1, T-2]   utils/misc.py:77-81, spectral_repr.py:326-328
 *   mode 4  angle only                                                   (Phase(unwrap=False))
 * bare != 0 (modes 1-3, real input in `phase`): the plain fdiff_* of utils/misc.py:65-81, no unwrap, no division.
 * Sequential in t per (clip, bin) column with torch's CPU arithmetic (double cumsum accumulator). */
int at_phase_scan(const float *X_complex, const float *phase, int64_t B, int64_t T, int64_t F, int mode, int bare,
                  const float *frame_window, const float *offset, const float *scale, float *out, void *stream);

/* IF.invert (spectral_repr.py:360-373): de-normalise (offset/scale may be NULL), undo the per-row scaling of
 * `method` (1 forward, 2 backward, 3 central; skipped when rescale == 0: the bare fint_* of utils/misc.py:83-104)
 * and integrate along t.  out != y. */
int at_phase_integrate(const float *y, int64_t B, int64_t T, int64_t F, int method, int rescale, const float *offset,
                       const float *scale, float *out, void *stream);

/* The same scan writing frames ld_out >= F floats apart: the phase half of a stacked (.., T, 2, F) representation is
 * filled in place (out = stacked + F, ld_out = 2 F) -- PolarIF.forward without the torch.stack copy. */
int at_phase_scan_strided(const float *X_complex, const float *phase, int64_t B, int64_t T, int64_t F, int mode, int bare,
                          const float *frame_window, const float *offset, const float *scale, float *out, int64_t ld_out,
                          void *stream);
/* PolarIF.forward in one pass (spectral_repr.py:505-537 = Magnitude.forward :186-200, :222-227 next to IF.forward :318-335,
 * :350-358, stacked on dim -2): X (B, T, F) complex64 -> out_stacked (B, T, 2, F) with [.., 0, :] =
 * normalise(contrast(|X| @ bank)) and [.., 1, :] = normalise(IF(X)) (`method`, frame_window, if_offset / if_scale as in
 * at_phase_scan).  The bank is given by filter (device arrays): band_start[f] first bin with a non-zero weight,
 * band_len[f] bins up to the last one (0: empty filter), band_off[f] offset of those weights in band_w (n_w floats;
 * start + len <= F, off + len <= n_w are the caller's to guarantee).  Both halves equal at_mel_project_banded's and
 * at_phase_scan_strided's bit for bit; the spectrum is read once.  One block per clip: AT_EUNSUPPORTED unless B >= 64,
 * 256 <= F <= 2048 (two-column clip blocks; the four-column form is not built for the magnitude side), rows that are not whole
 * 64-byte segments and a bank whose weights fit LDS beside eight rows --
 * the caller then runs those two entry points. */
int at_polarif_forward(const float *X_complex, int64_t B, int64_t T, int64_t F, int method, const float *frame_window,
                       const float *if_offset, const float *if_scale, const int *band_start, const int *band_len,
                       const int *band_off, const float *band_w, int64_t n_w, int contrast, const float *mag_offset,
                       const float *mag_scale, float eps, float *out_stacked, void *stream);
/* IF.invert fused with SpectralRepresentation.invert's mag * exp(i * phase) (spectral_repr.py:360-373, 449-451): y has
 * frames ld_y >= F floats apart (the phase half of a stacked tensor: y = stacked + F, ld_y = 2 F), mag is (B, T, F)
 * contiguous, out_complex (B, T, F) complex64. */
int at_phase_integrate_polar(const float *y, int64_t ld_y, int64_t B, int64_t T, int64_t F, int method, const float *offset,
                             const float *scale, const float *mag, float *out_complex, void *stream);

/* mag * exp(i * phase) -> complex64 (SpectralRepresentation.invert, spectral_repr.py:449-451). */
int at_polar_to_complex(const float *mag, const float *phase, int64_t n, float *out_complex, void *stream);

/* ---- sinebank inversion --------------------------------------------------- */
/* STFT/DGT.get_sinebank_inversion (stft.py:180-191): y[b, n] = sum_k env[b,k,n] sin(c_k t_n + phi_k), env = linear
 * interpolation of x / max|x| along time, / 2 pi.  x: (B, T, F) magnitudes; c[F] = fl(2 pi) f_k, t[L], phi[F]: the
 * reference's own fp32 frequency / time / random-phase vectors (host-computed with the same torch calls).  Per
 * 128-sample block cb of the output: block_frame_offset[p * nblocks + cb] (int64, elements) = F * (first frame the
 * block interpolates from + p), clamped to the last frame, p = 0..n_pass-1 (3 for hop >= 128); W[p * L + n] =
 * weight of that frame for sample n (zero where unused).  max_abs: device scalar max|x|.  out: (B, L), before the
 * final division by its max.  n_pass exact-fp32 MFMA contractions against one shared oscillator matrix + a
 * pointwise combine. */
size_t at_sinebank_workspace_bytes(int64_t B, int F, int64_t L, int n_pass);
int at_sinebank_offline(const float *x, int64_t B, int64_t T, int F, const float *c, const float *t, const float *phi,
                        int64_t L, int n_pass, const int64_t *block_frame_offset, const float *W, const float *max_abs,
                        float *out, void *workspace, size_t workspace_bytes, void *stream);

/* RealtimeSTFT/RealtimeDGT.get_sinebank_inversion (stft.py:276-291, dgt.py:356-371):
 * out[s, t, n] = (1/F) sum_k x[s, t, k] sin(c_k tau[t, n] + phi[s, k]);  x (S, T, F), tau (T, N), phi (S, F), out (S, T, N).
 * window_or_null (N floats): the frames are multiplied by it -- what invert(mode="sinebank") hands to OverlapAdd
 * (stft.py:303-304, dgt.py:321-322: get_sinebank_inversion(x) * inv_window). */
int at_sinebank_realtime(const float *x, int64_t S, int T, int F, int N, const float *c, const float *tau,
                         const float *phi, const float *window_or_null, float *out, void *stream);

/* ---- audio front end ------------------------------------------------------------------------------------- */
/* torchaudio.transforms.Resample(orig, new) with default arguments, as utils/misc.py:31-33 uses it (algorithm
 * restated, torchaudio is not in the reference tree).  x: (rows, L); orig/new: the rates divided by their gcd;
 * filters: new x (2*width + orig) float32 polyphase bank; out: (rows, out_len), out_len = ceil(new * L / orig). */
int at_resample_sinc(const float *x, int64_t rows, int64_t L, int orig, int new_, int width, const float *filters,
                     int64_t out_len, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ACIDS_HIP_H */
