// Dev micro-benchmark (not product), round 3: splits the STFT row access pattern of stream_pattern.hip into its parts.
//   LD: 0 none, 1 sliding window (2 x 8 B per lane and frame, 8 at a chunk start), 2 spectrum rows (8 x 8 B + 4 B)
//   ST: 0 none, 1 rows of 4104 B at their natural (8-byte aligned) base + one-lane Nyquist store,
//       2 the same bytes as ONE stream cut into 512-B aligned blocks (8 B per lane, 8 full stores per frame and a
//         ninth every 64 frames), 3 the same as 1-KB aligned blocks (16 B per lane), 4 audio hops (2 x 8 B per lane)
//   NT: non-temporal stores
// Frames are dealt to waves in chunks of G consecutive frames, chunk c of trip i going to wave (c mod waves):
// G = frames / waves is the product kernels' "one long run per wave", small G a compact advancing front.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LD, int ST, int NT>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, const float2* __restrict__ spec_in,
                                         float2* __restrict__ out, float* __restrict__ audio, long long total,
                                         long long G, long long nwaves, float* sink) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  vf2 raw[8];
  for (int m = 0; m < 8; ++m) raw[m] = (vf2){0.f, 0.f};
  vf2 accum = {0.f, 0.f};
  const long long nchunks = (total + G - 1) / G;
  for (long long c = w; c < nchunks; c += nwaves) {
    const long long f0 = c * G;
    long long f1 = f0 + G;
    if (f1 > total) f1 = total;
    if (LD == 1) {
      const vf2* src = reinterpret_cast<const vf2*>(x + f0 * 256);
      for (int m = 0; m < 6; ++m) raw[m + 2] = src[lane + 64 * m];
    }
    for (long long f = f0; f < f1; ++f) {
      if (LD == 1) {
        const vf2* src = reinterpret_cast<const vf2*>(x + f * 256 + 768);
        for (int m = 0; m < 6; ++m) raw[m] = raw[m + 2];
        raw[6] = src[lane];
        raw[7] = src[lane + 64];
      } else if (LD == 2) {
        const vf2* src = reinterpret_cast<const vf2*>(spec_in + f * 513);
        for (int m = 0; m < 8; ++m) raw[m] = src[lane + 64 * m];
        accum.x += reinterpret_cast<const float*>(src + 512)[0];
      }
      vf2 v[8];
      for (int m = 0; m < 8; ++m) v[m] = raw[m] * (vf2){1.0001f, 0.9999f};
      if (ST == 0) {
        for (int m = 0; m < 8; ++m) accum += v[m];
      } else if (ST == 1) {
        vf2* row = reinterpret_cast<vf2*>(out + f * 513);
        for (int m = 0; m < 8; ++m) {
          if (NT) __builtin_nontemporal_store(v[m], &row[lane + 64 * m]); else row[lane + 64 * m] = v[m];
        }
        if (lane == 0) row[512] = v[0];
      } else if (ST == 2) {
        // 512-B aligned blocks of the flat complex stream: blocks [f*513/64 .. (f+1)*513/64)
        const long long b0 = (f * 513) >> 6, b1 = ((f + 1) * 513) >> 6;
        vf2* base = reinterpret_cast<vf2*>(out);
        int m = 0;
        for (long long blk = b0; blk < b1; ++blk, ++m) {
          vf2* dst = base + blk * 64 + lane;
          if (NT) __builtin_nontemporal_store(v[m & 7], dst); else *dst = v[m & 7];
        }
      } else if (ST == 3) {
        // 1-KB aligned blocks (16 B per lane)
        const long long b0 = (f * 513) >> 7, b1 = ((f + 1) * 513) >> 7;
        vf4* base = reinterpret_cast<vf4*>(out);
        int m = 0;
        for (long long blk = b0; blk < b1; ++blk, ++m) {
          vf4 q = {v[(2 * m) & 7].x, v[(2 * m) & 7].y, v[(2 * m + 1) & 7].x, v[(2 * m + 1) & 7].y};
          vf4* dst = base + blk * 64 + lane;
          if (NT) __builtin_nontemporal_store(q, dst); else *dst = q;
        }
      } else if (ST == 4) {
        vf2* dst = reinterpret_cast<vf2*>(audio + f * 256);
        for (int m = 0; m < 2; ++m) {
          if (NT) __builtin_nontemporal_store(v[m] + v[m + 2] + v[m + 4] + v[m + 6], &dst[lane + 64 * m]);
          else dst[lane + 64 * m] = v[m] + v[m + 2] + v[m + 4] + v[m + 6];
        }
      }
    }
  }
  if (ST == 0 && accum.x + accum.y == 123.456f) sink[0] = accum.x;
}

static const long long B = 1024, L = 176400 + 2048, T = 690;
static float* x; static float2 *spec, *spec2; static float *audio, *sink;

template <int LD, int ST, int NT>
static void run(const char* name, long long G, int wpb, int bpc) {
  const long long total = B * T;
  const long long blocks = 256LL * bpc;
  const long long nwaves = blocks * wpb;
  if (G <= 0) G = (total + nwaves - 1) / nwaves;
  hipEvent_t ev_s, ev_e;
  CHECK(hipEventCreate(&ev_s)); CHECK(hipEventCreate(&ev_e));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((k<LD, ST, NT>), dim3(blocks), dim3(64 * wpb), 0, 0, x, spec2, spec, audio, total, G, nwaves, sink);
  CHECK(hipEventRecord(ev_s, 0));
  for (int i = 0; i < 10; ++i)
    hipLaunchKernelGGL((k<LD, ST, NT>), dim3(blocks), dim3(64 * wpb), 0, 0, x, spec2, spec, audio, total, G, nwaves, sink);
  CHECK(hipEventRecord(ev_e, 0)); CHECK(hipEventSynchronize(ev_e));
  float ms; CHECK(hipEventElapsedTime(&ms, ev_s, ev_e)); ms /= 10;
  const double bytes = (LD == 1 ? 1024.0 : LD == 2 ? 4104.0 : 0.0) + (ST == 0 ? 0.0 : ST == 4 ? 1024.0 : 4104.0);
  printf("%-40s G=%-5lld waves/CU=%-3d %.3f ms  %.2f TB/s\n", name, G, wpb * bpc, ms, total * bytes / ms / 1e9);
  fflush(stdout);
}

__global__ __launch_bounds__(256) void fill_k(vf4* __restrict__ out, long long n16, int nt) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const vf4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    if (nt) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}
static void run_fill(int bpc, int nt) {
  const long long n16 = B * T * 513 * 8 / 16;
  hipEvent_t ev_s, ev_e;
  CHECK(hipEventCreate(&ev_s)); CHECK(hipEventCreate(&ev_e));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_k, dim3(256 * bpc), dim3(256), 0, 0, (vf4*)spec, n16, nt);
  CHECK(hipEventRecord(ev_s, 0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill_k, dim3(256 * bpc), dim3(256), 0, 0, (vf4*)spec, n16, nt);
  CHECK(hipEventRecord(ev_e, 0)); CHECK(hipEventSynchronize(ev_e));
  float ms; CHECK(hipEventElapsedTime(&ms, ev_s, ev_e)); ms /= 10;
  printf("grid-stride fill 16 B/lane%s, %d blocks/CU: %.3f ms  %.2f TB/s\n", nt ? " nt" : "", bpc, ms, n16 * 16.0 / ms / 1e9);
}

int main(int argc, char** argv) {
  if (argc > 1) {     // second sweep: very short chunks (a compact write front) and the plain fill ceiling
    CHECK(hipMalloc(&x, B * L * 4)); CHECK(hipMalloc(&spec, B * T * 513 * 8 + 4096)); CHECK(hipMalloc(&spec2, B * T * 513 * 8 + 4096));
    CHECK(hipMalloc(&audio, B * L * 4)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(x, 0, B * L * 4)); CHECK(hipMemset(spec2, 0, B * T * 513 * 8 + 4096));
    for (int bpc : {2, 4, 8}) { run_fill(bpc, 0); run_fill(bpc, 1); }
    for (int occ : {16, 24, 32}) {
      const int wpb = 8, bpc = occ / 8;
      printf("---- %d waves per CU\n", occ);
      for (long long G : {1LL, 2LL, 4LL, 8LL, 32LL}) {
        run<0, 2, 0>("write flat 512-B aligned (8 B)", G, wpb, bpc);
        run<0, 2, 1>("write flat 512-B aligned nt", G, wpb, bpc);
        run<0, 3, 1>("write flat 1-KB aligned nt", G, wpb, bpc);
        run<1, 2, 0>("fwd: audio -> flat 512-B aligned", G, wpb, bpc);
        run<1, 2, 1>("fwd: audio -> flat 512-B aligned nt", G, wpb, bpc);
        run<1, 3, 1>("fwd: audio -> flat 1-KB aligned nt", G, wpb, bpc);
        run<2, 4, 0>("inv: rows -> audio", G, wpb, bpc);
        run<2, 4, 1>("inv: rows -> audio nt", G, wpb, bpc);
      }
    }
    return 0;
  }
  CHECK(hipMalloc(&x, B * L * 4)); CHECK(hipMalloc(&spec, B * T * 513 * 8 + 4096)); CHECK(hipMalloc(&spec2, B * T * 513 * 8 + 4096));
  CHECK(hipMalloc(&audio, B * L * 4)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(x, 0, B * L * 4)); CHECK(hipMemset(spec2, 0, B * T * 513 * 8 + 4096));
  const long long Gs[] = {0, 64, 16, 8, 4};
  for (int occ : {12, 16}) {
    const int wpb = 4, bpc = occ / 4;
    printf("---- %d waves per CU\n", occ);
    for (long long G : Gs) {
      run<1, 0, 0>("read audio only", G, wpb, bpc);
      run<0, 1, 0>("write rows only (8 B, misaligned)", G, wpb, bpc);
      run<0, 2, 0>("write flat 512-B aligned (8 B)", G, wpb, bpc);
      run<0, 3, 0>("write flat 1-KB aligned (16 B)", G, wpb, bpc);
      run<0, 2, 1>("write flat 512-B aligned nt", G, wpb, bpc);
      run<0, 3, 1>("write flat 1-KB aligned nt", G, wpb, bpc);
      run<1, 1, 0>("fwd: audio -> rows misaligned", G, wpb, bpc);
      run<1, 2, 0>("fwd: audio -> flat 512-B aligned", G, wpb, bpc);
      run<1, 3, 0>("fwd: audio -> flat 1-KB aligned", G, wpb, bpc);
      run<1, 3, 1>("fwd: audio -> flat 1-KB aligned nt", G, wpb, bpc);
      run<2, 0, 0>("read rows only", G, wpb, bpc);
      run<2, 4, 0>("inv: rows -> audio", G, wpb, bpc);
    }
  }
  for (int occ : {4, 8}) {
    const int wpb = 4, bpc = occ / 4;
    printf("---- %d waves per CU\n", occ);
    for (long long G : {0LL, 16LL}) {
      run<0, 1, 0>("write rows only (8 B, misaligned)", G, wpb, bpc);
      run<0, 3, 0>("write flat 1-KB aligned (16 B)", G, wpb, bpc);
      run<1, 1, 0>("fwd: audio -> rows misaligned", G, wpb, bpc);
      run<1, 3, 0>("fwd: audio -> flat 1-KB aligned", G, wpb, bpc);
      run<2, 4, 0>("inv: rows -> audio", G, wpb, bpc);
    }
  }
  return 0;
}
