// capi.hip -- extern "C" boundary of libacids_hip.so (see include/acids_hip.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <vector>

#include "../../include/acids_hip.h"
#include "band_bank.h"
#include "fft512.h"

#include <atomic>
#include "variants.h"
namespace at_hip {
// stft1024.hip
int launch_stft1024_fwd(const float*, long long, long long, long long, long long, int, int, const float*,
                        const float2*, float2*, float*, hipStream_t);
int launch_stft1024_h256_fwd(const float*, long long, long long, long long, long long, const float*, const float2*,
                             float2*, float*, const BandBank*, float*, const float*, const float*, float, int, int, int,
                             hipStream_t, const PolarOut* polar = nullptr, int hop = 256);
int launch_istft1024_ola(const float2*, const float*, const float*, long long, long long, int, const float*,
                         const float*, const float2*, float*, hipStream_t, const float2* gl_tprev = nullptr,
                         float gl_mom = 0.f);
int launch_irfft1024_frames(const float2*, const float*, const float*, long long, const float*, const float2*,
                            float*, hipStream_t);
// stft_generic.hip
int launch_rfft_generic(const float*, long long, long long, long long, long long, int, int, int, const float*,
                        float2*, float*, hipStream_t);
int launch_rfft_mixed(const float*, long long, long long, long long, long long, int, int, int, const float*, float2*,
                      float*, hipStream_t);
int launch_irfft_mixed(const float2*, const float*, const float*, long long, int, const float*, float*, hipStream_t);
int launch_irfft_generic(const float2*, const float*, const float*, long long, int, const float*, float*,
                         hipStream_t);
int launch_ola_gather(const float*, long long, long long, int, int, const float*, float*, hipStream_t);
// stft2048.hip
int launch_stft2048_fwd(const float*, long long, long long, long long, long long, int, int, const float*, const float2*,
                        const float2*, float2*, float*, hipStream_t);
int launch_irfft2048_frames(const float2*, const float*, const float*, long long, const float*, const float2*,
                            const float2*, float*, hipStream_t);
int launch_stft512_mel(const float*, long long, long long, long long, long long, int, const float*, const float2*,
                       const float2*, const BandBank*, int, int, const float*, const float*, float, float*, int, hipStream_t);
int launch_stft2048_mel(const float*, long long, long long, long long, long long, int, const float*, const float2*,
                        const float2*, const BandBank*, int, int, const float*, const float*, float, float*, int, hipStream_t);
int launch_istft2048_ola(const float2*, const float*, const float*, long long, long long, int, const float*,
                         const float*, const float2*, const float2*, float*, hipStream_t);
// stft4096.hip
int launch_stft4096_fwd(const float*, long long, long long, long long, long long, int, int, const float*, const float2*,
                        const float2*, float2*, float*, hipStream_t);
int launch_istft4096_ola(const float2*, const float*, const float*, long long, long long, int, const float*, const float*,
                         const float2*, const float2*, float*, hipStream_t);
int launch_irfft4096_frames(const float2*, const float*, const float*, long long, const float*, const float2*,
                            const float2*, float*, hipStream_t);
// stft_small.hip (n_fft 256 / 128: four / eight frames per wave-level FFT)
int launch_stft_small_fwd(int, const float*, long long, long long, long long, long long, int, int, const float*, const float2*,
                          const float2*, float2*, float*, hipStream_t);
int launch_irfft_small_frames(int, const float2*, const float*, const float*, long long, const float*, const float2*,
                              const float2*, float*, hipStream_t);
// stft512.hip
int launch_stft512_fwd(const float*, long long, long long, long long, long long, int, int, const float*, const float2*,
                       const float2*, float2*, float*, hipStream_t);
int launch_irfft512_frames(const float2*, const float*, const float*, long long, const float*, const float2*,
                           const float2*, float*, hipStream_t);
int launch_istft512_ola(const float2*, const float*, const float*, long long, long long, int, const float*, const float*,
                        const float2*, const float2*, float*, hipStream_t);

constexpr int kMaxDevices = 16;
static float2* g_twiddles[kMaxDevices] = {nullptr};
static float2* g_tw2048[kMaxDevices] = {nullptr};     // W2048^k, k = 0 .. 1023 (stft2048.hip), then W512^k, k = 0 .. 255 (stft512.hip)

static const float2* twiddles_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  return g_twiddles[dev];
}

static const float2* tw2048_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  return g_tw2048[dev];
}

static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
// every n_fft in [2, 16384] (odd sizes, transformed at full length, up to 8191: two LDS copies of the frame)
static bool fft_size_ok(int n) { return n >= 2 && n <= 16384 && (!(n & 1) || n < 8192); }
// sizes the mixed-radix kernels of stft_mixed.hip take (everything that is not a power of two >= 8)
static bool fft_mixed(int n) { return !is_pow2(n) || n < 8; }

__global__ void envelope_table_kernel(const float* w, int n_fft, int hop, int R, float* env) {
  // env[mask][r] = sum over q in mask (ascending) of w[hop*(R-1-q) + r]^2: the R = n_fft / hop frames that overlap a
  // hop, oldest first
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1 << R) * hop) return;
  int mask = i / hop, r = i - mask * hop;
  float s = 0.f;
  for (int q = 0; q < R; ++q)
    if (mask & (1 << q)) {
      int o = hop * (R - 1 - q) + r;
      if (o < n_fft) s += w[o] * w[o];
    }
  env[i] = s;
}
static std::atomic<int> g_variants[kVarCount];
int variant(int which) { return g_variants[which].load(std::memory_order_relaxed); }
}  // namespace at_hip

using namespace at_hip;

extern "C" {

// 2: at_sinebank_realtime takes the synthesis window; bf16 projection, at_oadd_push
// 3: any n_fft (odd sizes give torch.istft's hop (T-1) + 1 samples); Cartesian pack / unpack; strided phase scans
// 4: at_set_variant / at_get_variant (round 4; the library no longer reads environment variables)
int at_abi_version(void) { return 4; }

int at_set_variant(int which, int value) {
  if (which < 0 || which >= kVarCount || value < 0 || value > 4) return AT_EINVAL;
  g_variants[which].store(value, std::memory_order_relaxed);
  return AT_OK;
}

int at_get_variant(int which) {
  if (which < 0 || which >= kVarCount) return AT_EINVAL;
  return g_variants[which].load(std::memory_order_relaxed);
}

const char* at_error_string(int code) {
  switch (code) {
    case AT_OK: return "ok";
    case AT_EINVAL: return "invalid argument";
    case AT_EUNSUPPORTED: return "unsupported configuration";
    case AT_ENOTINIT: return "at_init() not called for this device";
    case AT_EWORKSPACE: return "workspace too small";
    case AT_ELAUNCH: return "HIP launch/runtime error";
    default: return "unknown error";
  }
}

int at_init(int device) {
  if (device < 0 || device >= kMaxDevices) return AT_EINVAL;
  if (g_twiddles[device]) return AT_OK;
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return AT_ELAUNCH;
  if (hipSetDevice(device) != hipSuccess) return AT_ELAUNCH;
  std::vector<float2> tab(kTwiddleCount);
  const double two_pi = 6.283185307179586476925286766559;
  for (int k = 1; k < 8; ++k)
    for (int l = 0; l < 64; ++l) {
      double a1 = -two_pi * (double)(l * k) / 512.0;
      tab[(k - 1) * 64 + l] = make_float2((float)cos(a1), (float)sin(a1));
      double a2 = -two_pi * (double)((l & 7) * k) / 64.0;
      tab[(7 + k - 1) * 64 + l] = make_float2((float)cos(a2), (float)sin(a2));
    }
  for (int m = 0; m < 8; ++m)
    for (int l = 0; l < 64; ++l) {
      double a = -two_pi * (double)(l + 64 * m) / 1024.0;
      tab[(14 + m) * 64 + l] = make_float2((float)cos(a), (float)sin(a));
    }
  float2* d = nullptr;
  int rc = AT_OK;
  // the table, then the tile-counter ring of the persistent forward kernels (stft1024.hip: tile_counter_slot), zeroed
  if (hipMalloc((void**)&d, sizeof(float2) * (kTwiddleCount + kTileCtrSlots)) != hipSuccess) rc = AT_ELAUNCH;
  if (rc == AT_OK && hipMemcpy(d, tab.data(), sizeof(float2) * kTwiddleCount, hipMemcpyHostToDevice) != hipSuccess)
    rc = AT_ELAUNCH;
  if (rc == AT_OK && hipMemset(d + kTwiddleCount, 0, sizeof(float2) * kTileCtrSlots) != hipSuccess) rc = AT_ELAUNCH;
  float2* d2 = nullptr;
  if (rc == AT_OK) {
    // [0, 1024): W2048^k; [1024, 1280): W512^k; [1280, 1664): W512^(r k), r = 1..3, k < 128 (n_fft 256);
    // [1664, 2112): W512^(r k), r = 1..7, k < 64 (n_fft 128)
    // [2112, 4160): W4096^k, k < 2048; [4160, 5696): W2048^(r k), r = 1..3, k < 512 (n_fft 4096, stft4096.hip)
    std::vector<float2> t2(1024 + 256 + 384 + 448 + 2048 + 1536);
    for (int k = 0; k < 2048; ++k) {
      const double a = -two_pi * (double)k / 4096.0;
      t2[2112 + k] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int r = 1; r < 4; ++r)
      for (int k = 0; k < 512; ++k) {
        const double a = -two_pi * (double)(r * k) / 2048.0;
        t2[4160 + (r - 1) * 512 + k] = make_float2((float)cos(a), (float)sin(a));
      }
    for (int r = 1; r < 4; ++r)
      for (int k = 0; k < 128; ++k) {
        const double a = -two_pi * (double)(r * k) / 512.0;
        t2[1280 + (r - 1) * 128 + k] = make_float2((float)cos(a), (float)sin(a));
      }
    for (int r = 1; r < 8; ++r)
      for (int k = 0; k < 64; ++k) {
        const double a = -two_pi * (double)(r * k) / 512.0;
        t2[1664 + (r - 1) * 64 + k] = make_float2((float)cos(a), (float)sin(a));
      }
    for (int k = 0; k < 1024; ++k) {
      const double a = -two_pi * (double)k / 2048.0;
      t2[k] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int k = 0; k < 256; ++k) {
      const double a = -two_pi * (double)k / 512.0;
      t2[1024 + k] = make_float2((float)cos(a), (float)sin(a));
    }
    if (hipMalloc((void**)&d2, sizeof(float2) * t2.size()) != hipSuccess) rc = AT_ELAUNCH;
    if (rc == AT_OK && hipMemcpy(d2, t2.data(), sizeof(float2) * t2.size(), hipMemcpyHostToDevice) != hipSuccess)
      rc = AT_ELAUNCH;
  }
  if (rc == AT_OK) {
    g_tw2048[device] = d2;
    g_twiddles[device] = d;
  }
  (void)hipSetDevice(prev);
  return rc;
}

int at_stft_forward(const float* x, int64_t B, int64_t L, int64_t clip_stride, int64_t T, int n_fft, int hop,
                    int center, const float* window, float* out_complex, float* phase, void* stream) {
  if (B < 0 || T < 0 || L < 0 || hop <= 0 || n_fft <= 0) return AT_EINVAL;
  if (B * T == 0) return AT_OK;
  if (!x || !window || !out_complex) return AT_EINVAL;
  if (!fft_size_ok(n_fft)) return AT_EUNSUPPORTED;
  if (center && L <= n_fft / 2) return AT_EINVAL;  // torch.stft: reflect pad must be < L
  hipStream_t s = (hipStream_t)stream;
  if (n_fft == 1024 && (((uintptr_t)window) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    if (!tw) return AT_ENOTINIT;
    if ((hop == 256 || hop == 128 || hop == 512) && center && (clip_stride & 1) == 0)
      return launch_stft1024_h256_fwd(x, B, L, clip_stride, T, window, tw, (float2*)out_complex, phase, nullptr,
                                      nullptr, nullptr, nullptr, 0.f, 0, 0, 0, s, nullptr, hop);
    return launch_stft1024_fwd(x, B, L, clip_stride, T, hop, center, window, tw, (float2*)out_complex, phase, s);
  }
  if (n_fft == 2048 && (((uintptr_t)window) & 15) == 0) {      // two 512-point register FFTs + a radix-2 stage per frame
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_stft2048_fwd(x, B, L, clip_stride, T, hop, center, window, tw, tw2k, (float2*)out_complex, phase, s);
  }
  if (n_fft == 4096 && (((uintptr_t)window) & 15) == 0) {      // four 512-point register FFTs + a radix-4 stage per frame
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_stft4096_fwd(x, B, L, clip_stride, T, hop, center, window, tw, tw2k + 2112, (float2*)out_complex, phase, s);
  }
  if ((n_fft == 256 || n_fft == 128) && (((uintptr_t)window) & 7) == 0) {     // four / eight frames per register FFT
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_stft_small_fwd(n_fft, x, B, L, clip_stride, T, hop, center, window, tw, tw2k + (n_fft == 256 ? 1280 : 1664),
                                 (float2*)out_complex, phase, s);
  }
  if (n_fft == 512 && (((uintptr_t)window) & 7) == 0) {        // two frames per 512-point register FFT
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_stft512_fwd(x, B, L, clip_stride, T, hop, center, window, tw, tw2k + 1024, (float2*)out_complex, phase, s);
  }
  if (fft_mixed(n_fft))
    return launch_rfft_mixed(x, B, L, clip_stride, T, n_fft, hop, center, window, (float2*)out_complex, phase, s);
  return launch_rfft_generic(x, B, L, clip_stride, T, n_fft, hop, center, window, (float2*)out_complex, phase, s);
}

int at_stft_mel_forward(const float* x, int64_t B, int64_t L, int64_t clip_stride, int64_t T, int n_fft, int hop,
                        const float* window, const int32_t* lane_filter, const int32_t* lane_start,
                        const float* band_weights, int n_filters, int n_passes, const int32_t* pass_len_host,
                        int contrast, int power2, const float* offset, const float* scale, float eps,
                        float* out_complex_or_null, float* phase_or_null, float* feat, int feat_channel_major,
                        void* stream) {
  if (B < 0 || T < 0 || L < 0) return AT_EINVAL;
  // features only at n_fft 2048 / 512 (stft2048.hip, stft512.hip)
  const bool feat_only_2048 = (n_fft == 2048 || n_fft == 512) && hop >= 1 && !out_complex_or_null && !phase_or_null;
  if (!feat_only_2048) {
    if (n_fft != 1024 || (hop != 256 && hop != 128 && hop != 512) || (clip_stride & 1)) return AT_EUNSUPPORTED;
    if (hop != 256 && feat_channel_major) return AT_EUNSUPPORTED;
  }
  if (B * T == 0) return AT_OK;
  if (!x || !window || !feat || !lane_filter || !lane_start || !band_weights || !pass_len_host) return AT_EINVAL;
  if (n_filters <= 0 || n_passes <= 0 || n_passes > 16 || n_filters > 64 * n_passes) return AT_EINVAL;
  if ((offset == nullptr) != (scale == nullptr)) return AT_EINVAL;
  if (L <= n_fft / 2 || (((uintptr_t)window) & (n_fft == 2048 ? 15 : 7)) || (((uintptr_t)band_weights) & 15)) return AT_EINVAL;
  const float2* tw = twiddles_for_current_device();
  if (!tw) return AT_ENOTINIT;
  BandBank bank = {lane_filter, lane_start, band_weights, n_filters, n_passes, {0}};
  long long table_floats = 0;
  for (int q = 0; q < n_passes; ++q) {
    if (pass_len_host[q] < 0 || pass_len_host[q] > 128 || (pass_len_host[q] & 3)) return AT_EINVAL;  // 4 bins per step
    bank.pass_len[q] = pass_len_host[q];
    table_floats += 64LL * pass_len_host[q];
  }
  if (table_floats > 8192) return AT_EUNSUPPORTED;   // LDS copy of the band weights
  if (feat_only_2048) {
    const float2* tw2k = tw2048_for_current_device();
    if (!tw2k) return AT_ENOTINIT;
    if (n_fft == 512)
      return launch_stft512_mel(x, B, L, clip_stride, T, hop, window, tw, tw2k + 1024, &bank, contrast, power2, offset, scale, eps,
                                feat, feat_channel_major, (hipStream_t)stream);
    return launch_stft2048_mel(x, B, L, clip_stride, T, hop, window, tw, tw2k, &bank, contrast, power2, offset, scale, eps, feat,
                               feat_channel_major, (hipStream_t)stream);
  }
  return launch_stft1024_h256_fwd(x, B, L, clip_stride, T, window, tw, (float2*)out_complex_or_null, phase_or_null, &bank,
                                  feat, offset, scale, eps, contrast, power2, feat_channel_major, (hipStream_t)stream,
                                  nullptr, hop);
}

int at_stft_polar_forward(const float* x, int64_t B, int64_t L, int64_t clip_stride, int64_t T, int n_fft, int hop,
                          const float* window, const int32_t* lane_filter, const int32_t* lane_start,
                          const float* band_weights, int n_filters, int n_passes, const int32_t* pass_len_host,
                          int contrast, const float* mag_offset, const float* mag_scale, float eps,
                          const float* phase_offset, const float* phase_scale, float* out_stacked, void* stream) {
  if (B < 0 || T < 0 || L < 0) return AT_EINVAL;
  if (n_fft != 1024 || hop != 256 || (clip_stride & 1)) return AT_EUNSUPPORTED;
  if (B * T == 0) return AT_OK;
  if (!x || !window || !out_stacked || !lane_filter || !lane_start || !band_weights || !pass_len_host) return AT_EINVAL;
  const int F = n_fft / 2 + 1;
  if (n_filters != F || n_passes <= 0 || n_passes > 16 || n_filters > 64 * n_passes) return AT_EINVAL;   // stacked halves
  if ((mag_offset == nullptr) != (mag_scale == nullptr) || (phase_offset == nullptr) != (phase_scale == nullptr))
    return AT_EINVAL;
  if (L <= n_fft / 2 || (((uintptr_t)window) & 7) || (((uintptr_t)band_weights) & 15)) return AT_EINVAL;
  const float2* tw = twiddles_for_current_device();
  if (!tw) return AT_ENOTINIT;
  BandBank bank = {lane_filter, lane_start, band_weights, n_filters, n_passes, {0}};
  long long table_floats = 0;
  for (int q = 0; q < n_passes; ++q) {
    if (pass_len_host[q] < 0 || pass_len_host[q] > 128 || (pass_len_host[q] & 3)) return AT_EINVAL;
    bank.pass_len[q] = pass_len_host[q];
    table_floats += 64LL * pass_len_host[q];
  }
  if (table_floats > 8192) return AT_EUNSUPPORTED;
  const PolarOut polar = {out_stacked + F, 2LL * F, 2LL * F, phase_offset, phase_scale};
  return launch_stft1024_h256_fwd(x, B, L, clip_stride, T, window, tw, nullptr, nullptr, &bank, out_stacked, mag_offset,
                                  mag_scale, eps, contrast, 0, 0, (hipStream_t)stream, &polar);
}

int at_istft_envelope_table(const float* inv_window, int n_fft, int hop, float* env16, void* stream) {
  if (!inv_window || !env16 || hop <= 0 || n_fft <= 0) return AT_EINVAL;
  if (n_fft % hop) return AT_EUNSUPPORTED;
  const int R = n_fft / hop;
  if (R != 2 && R != 4 && R != 8) return AT_EUNSUPPORTED;
  int total = (1 << R) * hop;
  hipLaunchKernelGGL(envelope_table_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, inv_window,
                     n_fft, hop, R, env16);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

static bool istft_fast(int n_fft, int hop, const float* env16, const float* w) {
  return n_fft == 1024 && (hop == 128 || hop == 256 || hop == 512) && env16 != nullptr && (((uintptr_t)w) & 7) == 0 &&
         (((uintptr_t)env16) & 7) == 0;
}

static bool istft2048_fused(int n_fft, int hop, const float* env, const float* w, const float* y) {
  return n_fft == 2048 && (hop == 256 || hop == 512 || hop == 1024) && env != nullptr && (((uintptr_t)w) & 15) == 0 &&
         (((uintptr_t)env) & 15) == 0 && (((uintptr_t)y) & 15) == 0;
}

size_t at_istft_workspace_bytes(int64_t B, int64_t T, int n_fft, int hop) {
  if (n_fft == 1024 && (hop == 128 || hop == 256 || hop == 512)) return 0;   // with the envelope table; see at_istft
  if (n_fft == 2048 && (hop == 256 || hop == 512 || hop == 1024)) return 0;  // likewise (stft2048.hip)
  if (n_fft == 512 && (hop == 64 || hop == 128 || hop == 256)) return 0;      // likewise (stft512.hip)
  if (n_fft == 4096 && (hop == 512 || hop == 1024 || hop == 2048)) return 0;  // likewise (stft4096.hip)
  return (size_t)B * (size_t)T * (size_t)n_fft * sizeof(float);
}

int at_istft(const float* X_complex, const float* mag, const float* phase, int64_t B, int64_t T, int n_fft, int hop,
             const float* inv_window, const float* env16, float* y, void* workspace, size_t workspace_bytes,
             void* stream) {
  if (B < 0 || T < 0 || hop <= 0 || n_fft <= 0) return AT_EINVAL;
  if (B == 0 || T == 0 || (T == 1 && !(n_fft & 1))) return AT_OK;     // hop * (T - 1) + (n_fft & 1) samples per clip
  if (!inv_window || !y) return AT_EINVAL;
  if (!X_complex && !(mag && phase)) return AT_EINVAL;
  if (!fft_size_ok(n_fft)) return AT_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (istft_fast(n_fft, hop, env16, inv_window) && (((uintptr_t)y) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    if (!tw) return AT_ENOTINIT;
    return launch_istft1024_ola((const float2*)X_complex, mag, phase, B, T, hop, inv_window, env16, tw, y, s);
  }
  if (istft2048_fused(n_fft, hop, env16, inv_window, y)) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_istft2048_ola((const float2*)X_complex, mag, phase, B, T, hop, inv_window, env16, tw, tw2k, y, s);
  }
  if (n_fft == 4096 && (hop == 512 || hop == 1024 || hop == 2048) && env16 && (((uintptr_t)inv_window) & 15) == 0 &&
      (((uintptr_t)env16) & 15) == 0 && (((uintptr_t)y) & 15) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_istft4096_ola((const float2*)X_complex, mag, phase, B, T, hop, inv_window, env16, tw, tw2k + 2112, y, s);
  }
  if (n_fft == 512 && (hop == 64 || hop == 128 || hop == 256) && env16 && (((uintptr_t)inv_window) & 7) == 0 &&
      (((uintptr_t)env16) & 7) == 0 && (((uintptr_t)y) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_istft512_ola((const float2*)X_complex, mag, phase, B, T, hop, inv_window, env16, tw, tw2k + 1024, y, s);
  }
  size_t need = (size_t)B * (size_t)T * (size_t)n_fft * sizeof(float);
  if (!workspace || workspace_bytes < need) return AT_EWORKSPACE;
  int rc;
  if (n_fft == 2048 && (((uintptr_t)inv_window) & 15) == 0 && (((uintptr_t)workspace) & 15) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    rc = launch_irfft2048_frames((const float2*)X_complex, mag, phase, B * T, inv_window, tw, tw2k, (float*)workspace, s);
  } else if (n_fft == 4096 && (((uintptr_t)inv_window) & 15) == 0 && (((uintptr_t)workspace) & 15) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    rc = launch_irfft4096_frames((const float2*)X_complex, mag, phase, B * T, inv_window, tw, tw2k + 2112, (float*)workspace, s);
  } else if ((n_fft == 256 || n_fft == 128) && (((uintptr_t)inv_window) & 7) == 0 && (((uintptr_t)workspace) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    rc = launch_irfft_small_frames(n_fft, (const float2*)X_complex, mag, phase, B * T, inv_window, tw,
                                   tw2k + (n_fft == 256 ? 1280 : 1664), (float*)workspace, s);
  } else if (n_fft == 512 && (((uintptr_t)inv_window) & 7) == 0 && (((uintptr_t)workspace) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    rc = launch_irfft512_frames((const float2*)X_complex, mag, phase, B * T, inv_window, tw, tw2k + 1024, (float*)workspace, s);
  } else if (fft_mixed(n_fft)) {
    rc = launch_irfft_mixed((const float2*)X_complex, mag, phase, B * T, n_fft, inv_window, (float*)workspace, s);
  } else {
    rc = launch_irfft_generic((const float2*)X_complex, mag, phase, B * T, n_fft, inv_window, (float*)workspace, s);
  }
  if (rc) return rc;
  return launch_ola_gather((const float*)workspace, B, T, n_fft, hop, inv_window, y, s);
}

int at_istft_griffinlim(const float* mag, const float* rebuilt_complex, const float* tprev_complex_or_null,
                        float momentum_over_1p, int64_t B, int64_t T, int n_fft, int hop, const float* inv_window,
                        const float* env16, float* y, void* stream) {
  if (B < 0 || T < 0 || hop <= 0 || n_fft <= 0) return AT_EINVAL;
  if (B == 0 || T <= 1) return AT_OK;
  if (!mag || !rebuilt_complex || !inv_window || !y) return AT_EINVAL;
  if (!istft_fast(n_fft, hop, env16, inv_window) || (((uintptr_t)y) & 7)) return AT_EUNSUPPORTED;
  const float2* tw = twiddles_for_current_device();
  if (!tw) return AT_ENOTINIT;
  // first iteration (no previous spectrum): momentum 0 against the rebuilt spectrum itself
  const float2* tprev = tprev_complex_or_null ? (const float2*)tprev_complex_or_null : (const float2*)rebuilt_complex;
  const float mom = tprev_complex_or_null ? momentum_over_1p : 0.0f;
  return launch_istft1024_ola((const float2*)rebuilt_complex, mag, nullptr, B, T, hop, inv_window, env16, tw, y,
                              (hipStream_t)stream, tprev, mom);
}

int at_irfft_frames(const float* X_complex, const float* mag, const float* phase, int64_t nframes, int n_fft,
                    const float* inv_window, float* frames, void* stream) {
  if (nframes < 0 || n_fft <= 0) return AT_EINVAL;
  if (nframes == 0) return AT_OK;
  if (!inv_window || !frames) return AT_EINVAL;
  if (!X_complex && !(mag && phase)) return AT_EINVAL;
  if (!fft_size_ok(n_fft)) return AT_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (n_fft == 1024 && (((uintptr_t)inv_window) & 7) == 0 && (((uintptr_t)frames) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    if (!tw) return AT_ENOTINIT;
    return launch_irfft1024_frames((const float2*)X_complex, mag, phase, nframes, inv_window, tw, frames, s);
  }
  if (n_fft == 2048 && (((uintptr_t)inv_window) & 15) == 0 && (((uintptr_t)frames) & 15) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_irfft2048_frames((const float2*)X_complex, mag, phase, nframes, inv_window, tw, tw2k, frames, s);
  }
  if (n_fft == 4096 && (((uintptr_t)inv_window) & 15) == 0 && (((uintptr_t)frames) & 15) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_irfft4096_frames((const float2*)X_complex, mag, phase, nframes, inv_window, tw, tw2k + 2112, frames, s);
  }
  if ((n_fft == 256 || n_fft == 128) && (((uintptr_t)inv_window) & 7) == 0 && (((uintptr_t)frames) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_irfft_small_frames(n_fft, (const float2*)X_complex, mag, phase, nframes, inv_window, tw,
                                     tw2k + (n_fft == 256 ? 1280 : 1664), frames, s);
  }
  if (n_fft == 512 && (((uintptr_t)inv_window) & 7) == 0 && (((uintptr_t)frames) & 7) == 0) {
    const float2* tw = twiddles_for_current_device();
    const float2* tw2k = tw2048_for_current_device();
    if (!tw || !tw2k) return AT_ENOTINIT;
    return launch_irfft512_frames((const float2*)X_complex, mag, phase, nframes, inv_window, tw, tw2k + 1024, frames, s);
  }
  if (fft_mixed(n_fft))
    return launch_irfft_mixed((const float2*)X_complex, mag, phase, nframes, n_fft, inv_window, frames, s);
  return launch_irfft_generic((const float2*)X_complex, mag, phase, nframes, n_fft, inv_window, frames, s);
}

}  // extern "C"
