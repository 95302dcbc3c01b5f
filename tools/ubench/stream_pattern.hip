// Dev micro-benchmark (not product): which global access shapes reach the HBM rate for the STFT row pattern?
// Rows of 513 complex64 (4104 B) written per frame; input read as overlapping 1024-float frames, hop 256.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LOADS, int STORE16, int NT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float2* __restrict__ out, long long T, long long L,
                                         long long total, long long fpb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long f0 = (long long)blockIdx.x * fpb, f1 = f0 + fpb;
  if (f1 > total) f1 = total;
  float2 raw[8];
  for (int m = 0; m < 8; ++m) raw[m] = make_float2(0.f, 0.f);
  // LOADS==8: wave takes frames f0+wave, +4, ... ; LOADS==2: wave takes a contiguous quarter of the block's frames
  long long per = (f1 - f0 + 3) / 4;
  long long a = (LOADS == 8) ? f0 + wave : f0 + wave * per;
  long long bnd = (LOADS == 8) ? f1 : ((a + per < f1) ? a + per : f1);
  long long step = (LOADS == 8) ? 4 : 1;
  bool first = true;
  for (long long f = a; f < bnd; f += step) {
    long long b = f / T, t = f - b * T;
    long long start = t * 256 - 512;
    if (start < 0) start = 0;
    if (start + 1024 > L) start = L - 1024;
    const float2* src = reinterpret_cast<const float2*>(x + b * L + start);
    if (LOADS == 8 || first) {
      for (int m = 0; m < 8; ++m) raw[m] = src[lane + 64 * m];
      first = false;
    } else {
      for (int m = 0; m < 6; ++m) raw[m] = raw[m + 2];
      raw[6] = src[lane + 64 * 6];
      raw[7] = src[lane + 64 * 7];
    }
    float2 v[8];
    for (int m = 0; m < 8; ++m) v[m] = make_float2(raw[m].x * 1.0001f, raw[m].y * 0.9999f);
    float2* row = out + f * 513;
    if (STORE16) {
      // pretend lane holds bins (2 lane, 2 lane + 1) + 128 m: 16-B stores, 1 KB contiguous per instruction
      for (int m = 0; m < 4; ++m) {
        vf4 w = {v[2 * m].x, v[2 * m].y, v[2 * m + 1].x, v[2 * m + 1].y};
        vf4* dst = reinterpret_cast<vf4*>(reinterpret_cast<char*>(row) + (size_t)(2 * lane + 128 * m) * 8);
        if (NT) __builtin_nontemporal_store(w, dst); else *dst = w;
      }
      if (lane == 0) row[512] = v[0];
    } else {
      for (int m = 0; m < 8; ++m) {
        vf2 w2 = {v[m].x, v[m].y};
        if (NT) __builtin_nontemporal_store(w2, reinterpret_cast<vf2*>(&row[lane + 64 * m])); else row[lane + 64 * m] = v[m];
      }
      if (lane == 0) row[512] = v[0];
    }
  }
}

template <int LOADS, int STORE16, int NT>
static void run(const char* name, const float* x, float2* out, long long B, long long T, long long L, int bpc) {
  long long total = B * T;
  long long blocks = 256LL * bpc;
  long long fpb = (total + blocks - 1) / blocks;
  fpb = (fpb + 3) / 4 * 4;
  blocks = (total + fpb - 1) / fpb;
  hipEvent_t ev_s, ev_e;
  CHECK(hipEventCreate(&ev_s)); CHECK(hipEventCreate(&ev_e));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LOADS, STORE16, NT>), dim3(blocks), dim3(256), 0, 0, x, out, T, L, total, fpb);
  CHECK(hipEventRecord(ev_s, 0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<LOADS, STORE16, NT>), dim3(blocks), dim3(256), 0, 0, x, out, T, L, total, fpb);
  CHECK(hipEventRecord(ev_e, 0)); CHECK(hipEventSynchronize(ev_e));
  float ms; CHECK(hipEventElapsedTime(&ms, ev_s, ev_e)); ms /= 10;
  printf("%-34s bpc=%d  %.3f ms  %.2f TB/s\n", name, bpc, ms, total * 5128.0 / ms / 1e9);
}

int main() {
  const long long B = 1024, L = 176400, T = 690;
  float* x; float2* out;
  CHECK(hipMalloc(&x, B * L * 4)); CHECK(hipMalloc(&out, B * T * 513 * 8 + 1024));
  CHECK(hipMemset(x, 0, B * L * 4));
  for (int bpc : {4, 8}) {
    run<8, 0, 0>("8 loads(8B), 8B stores", x, out, B, T, L, bpc);
    run<2, 0, 0>("2 loads(8B) sliding, 8B stores", x, out, B, T, L, bpc);
    run<2, 1, 0>("2 loads, 16B stores", x, out, B, T, L, bpc);
    run<2, 0, 1>("2 loads, 8B nt stores", x, out, B, T, L, bpc);
    run<2, 1, 1>("2 loads, 16B nt stores", x, out, B, T, L, bpc);
    run<8, 1, 0>("8 loads, 16B stores", x, out, B, T, L, bpc);
  }
  return 0;
}
