// Dev micro-benchmark (not product), round 4: LAUNCH SHAPE of the n_fft-1024 access patterns.
// Round 3's harness (stream_pattern2.hip) was persistent everywhere and topped out at 5.5 TB/s on pure writes where
// torch's fill_ reaches 6.85 TB/s on the same box.  Here every pattern comes in two launch shapes:
//   P  persistent: 256 x bpc workgroups, unit u of trip i goes to wave (u mod nwaves)   [round 3's shape]
//   D  dispatch order: one workgroup per tile of (waves per block x G) consecutive units, as many workgroups as tiles,
//      handed out by the hardware dispatcher in blockIdx order (what an elementwise torch kernel does)
// Patterns:
//   fill   : 16 B / 8 B per lane, plain / nt
//   fwd    : audio (sliding window: 2 x 8 B per lane and frame, 6 more at a run start) -> the (B, T, 513) complex stream
//            written as 512-byte aligned blocks of 8 B per lane (the product kernels' store shape) [+ 512 B of features]
//   inv    : rows (8 x 8 B per lane + Nyquist) -> audio hops (2 x 8 B per lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static const long long B = 1024, L = 176400 + 2048, T = 690;
__device__ int g_think = 0;
__device__ int g_wide = 0;       // 1: the spectrum leaves as 1-KB blocks of 16 B per lane instead of 512-B blocks of 8 B      // s_sleep units (64 clocks) of "compute" per frame

static float* x; static float2 *spec, *spec2; static float *audio, *feat, *sink;

// ---------------------------------------------------------------- fills
template <int W16, int NT>
__global__ __launch_bounds__(256) void fill_tile_k(float* __restrict__ out, long long nbytes, int tile_bytes) {
  // one workgroup = one contiguous tile
  const long long base = (long long)blockIdx.x * tile_bytes;
  const int per_pass = 256 * (W16 ? 16 : 8);
  for (int off = threadIdx.x * (W16 ? 16 : 8); off < tile_bytes; off += per_pass) {
    const long long a = base + off;
    if (a >= nbytes) break;
    if (W16) {
      const vf4 v = {1.f, 2.f, 3.f, 4.f};
      vf4* p = reinterpret_cast<vf4*>(reinterpret_cast<char*>(out) + a);
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    } else {
      const vf2 v = {1.f, 2.f};
      vf2* p = reinterpret_cast<vf2*>(reinterpret_cast<char*>(out) + a);
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
}
template <int W16, int NT>
__global__ __launch_bounds__(256) void fill_stride_k(float* __restrict__ out, long long nbytes) {
  const long long stride = (long long)gridDim.x * blockDim.x * (W16 ? 16 : 8);
  for (long long a = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * (W16 ? 16 : 8); a < nbytes; a += stride) {
    if (W16) {
      const vf4 v = {1.f, 2.f, 3.f, 4.f};
      vf4* p = reinterpret_cast<vf4*>(reinterpret_cast<char*>(out) + a);
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    } else {
      const vf2 v = {1.f, 2.f};
      vf2* p = reinterpret_cast<vf2*>(reinterpret_cast<char*>(out) + a);
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
}

template <typename F>
static float time_ms(F launch, int warm = 3, int n = 10) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < warm; ++i) launch();
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < n; ++i) launch();
  CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipGetLastError());
  return ms / n;
}

static void fills() {
  const long long nbytes = B * T * 513 * 8;   // 2.9 GB, multiple of 16? 1024*690*513*8 yes
  printf("== fills of %.2f GB\n", nbytes / 1e9);
  for (int tile_kb : {4, 8, 16, 32, 64, 256}) {
    const int tb = tile_kb * 1024;
    const long long blocks = (nbytes + tb - 1) / tb;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL((fill_tile_k<1, 0>), dim3(blocks), dim3(256), 0, 0, (float*)spec, nbytes, tb); });
    printf("fill D tile %3d KB 16B plain : %.3f ms %.2f TB/s\n", tile_kb, ms, nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((fill_tile_k<1, 1>), dim3(blocks), dim3(256), 0, 0, (float*)spec, nbytes, tb); });
    printf("fill D tile %3d KB 16B nt    : %.3f ms %.2f TB/s\n", tile_kb, ms, nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((fill_tile_k<0, 0>), dim3(blocks), dim3(256), 0, 0, (float*)spec, nbytes, tb); });
    printf("fill D tile %3d KB  8B plain : %.3f ms %.2f TB/s\n", tile_kb, ms, nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((fill_tile_k<0, 1>), dim3(blocks), dim3(256), 0, 0, (float*)spec, nbytes, tb); });
    printf("fill D tile %3d KB  8B nt    : %.3f ms %.2f TB/s\n", tile_kb, ms, nbytes / ms / 1e9);
    fflush(stdout);
  }
  for (int bpc : {2, 4, 8}) {
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL((fill_stride_k<1, 0>), dim3(256 * bpc), dim3(256), 0, 0, (float*)spec, nbytes); });
    printf("fill P grid-stride %d blocks/CU 16B plain : %.3f ms %.2f TB/s\n", bpc, ms, nbytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((fill_stride_k<0, 0>), dim3(256 * bpc), dim3(256), 0, 0, (float*)spec, nbytes); });
    printf("fill P grid-stride %d blocks/CU  8B plain : %.3f ms %.2f TB/s\n", bpc, ms, nbytes / ms / 1e9);
  }

  {
    float ms = time_ms([&] { CHECK(hipMemsetAsync(spec, 0, nbytes, 0)); });
    printf("hipMemsetAsync                         : %.3f ms %.2f TB/s\n", ms, nbytes / ms / 1e9);
  }
  fflush(stdout);
}

// ---------------------------------------------------------------- forward / inverse access shapes
// One run = G consecutive frames [f0, f1) walked by one wave.
template <int NT, int FEAT>
__device__ __forceinline__ void fwd_run(const float* __restrict__ x, float2* __restrict__ out, float* __restrict__ feat,
                                        long long f0, long long f1, int lane) {
  vf2 raw[8];
  {
    const vf2* src = reinterpret_cast<const vf2*>(x + f0 * 256);
#pragma unroll
    for (int m = 0; m < 6; ++m) raw[m + 2] = src[lane + 64 * m];
  }
  vf2* base = reinterpret_cast<vf2*>(out);
  for (long long f = f0; f < f1; ++f) {
    const vf2* src = reinterpret_cast<const vf2*>(x + f * 256 + 768);
#pragma unroll
    for (int m = 0; m < 6; ++m) raw[m] = raw[m + 2];
    raw[6] = src[lane];
    raw[7] = src[lane + 64];
    vf2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = raw[m] * (vf2){1.0001f, 0.9999f};
    for (int q = g_think; q > 0; q -= 8) __builtin_amdgcn_s_sleep(8);      // stands in for the FFT + epilogue
    if (g_wide) {
      const long long b0 = (f * 513) >> 7, b1 = ((f + 1) * 513) >> 7;
      int m = 0;
      for (long long blk = b0; blk < b1; ++blk, ++m) {
        vf4 q = {v[(2 * m) & 7].x, v[(2 * m) & 7].y, v[(2 * m + 1) & 7].x, v[(2 * m + 1) & 7].y};
        vf4* dst = reinterpret_cast<vf4*>(out) + blk * 64 + lane;
        if (NT) __builtin_nontemporal_store(q, dst); else *dst = q;
      }
    } else {
    const long long b0 = (f * 513) >> 6, b1 = ((f + 1) * 513) >> 6;
    int m = 0;
    for (long long blk = b0; blk < b1; ++blk, ++m) {
      vf2* dst = base + blk * 64 + lane;
      if (NT) __builtin_nontemporal_store(v[m & 7], dst); else *dst = v[m & 7];
    }
    }
    if (FEAT) {
      vf2* fd = reinterpret_cast<vf2*>(feat + f * 128);
      fd[lane] = v[0] + v[1];
    }
  }
}
template <int NT>
__device__ __forceinline__ void inv_run(const float2* __restrict__ in, float* __restrict__ audio, long long f0, long long f1,
                                        int lane) {
  vf2 acc[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) acc[m] = (vf2){0.f, 0.f};
  for (long long f = f0; f < f1; ++f) {
    const vf2* src = reinterpret_cast<const vf2*>(in + f * 513);
    vf2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = __builtin_nontemporal_load(&src[lane + 64 * m]);
    const float ny = reinterpret_cast<const float*>(src + 512)[0];
    vf2 o0 = acc[0] + v[0] + (vf2){ny, ny}, o1 = acc[1] + v[1];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = acc[m + 2] + v[m + 2];
    acc[4] = v[6]; acc[5] = v[7];
    vf2* dst = reinterpret_cast<vf2*>(audio + f * 256);
    if (NT) { __builtin_nontemporal_store(o0, &dst[lane]); __builtin_nontemporal_store(o1, &dst[lane + 64]); }
    else { dst[lane] = o0; dst[lane + 64] = o1; }
  }
}

// KIND 0 fwd, 1 fwd + features, 2 inverse.
// SHAPE 0 persistent round-robin, 1 dispatch order (one tile per workgroup), 2 persistent with an atomic run counter
//       (a wave takes the next run in address order when it is free; the next index is requested one run ahead),
//       3 dispatch order with blocks re-mapped so that each XCD (blockIdx % 8) walks its own eighth of the stream
// lds_fill > 0: the workgroup first copies lds_fill bytes of tables from global memory into LDS and syncs (the product
// kernels' prologue); dynamic LDS also limits the occupancy like the product kernels' footprint does.
extern __shared__ float dyn_lds[];
template <int KIND, int SHAPE, int NT>
__global__ __launch_bounds__(512) void pat_k(const float* __restrict__ x, const float2* __restrict__ spec_in,
                                             float2* __restrict__ out, float* __restrict__ audio, float* __restrict__ feat,
                                             long long total, long long G, long long nwaves, int lds_fill,
                                             const float* __restrict__ tables, unsigned* __restrict__ counter,
                                             float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  if (lds_fill > 0) {
    for (int i = threadIdx.x * 4; i < lds_fill / 4; i += blockDim.x * 4)
      *reinterpret_cast<vf4*>(&dyn_lds[i]) = *reinterpret_cast<const vf4*>(&tables[i]);
    __syncthreads();
    if (dyn_lds[lane] == 123.f) sink[0] = 1.f;
  }
  const long long nruns = (total + G - 1) / G;
  if (SHAPE == 1 || SHAPE == 3) {
    long long blk = blockIdx.x;
    if (SHAPE == 3) {
      const long long per = (gridDim.x + 7) / 8;
      blk = (blk & 7) * per + (blk >> 3);
    }
    const long long w = __builtin_amdgcn_readfirstlane((int)(blk * wpb + (threadIdx.x >> 6)));
    if (w >= nruns) return;
    const long long f0 = w * G;
    long long f1 = f0 + G; if (f1 > total) f1 = total;
    if (KIND == 2) inv_run<NT>(spec_in, audio, f0, f1, lane);
    else fwd_run<NT, KIND == 1>(x, out, feat, f0, f1, lane);
  } else if (SHAPE == 5) {
    // persistent workgroups, tiles of wpb x G frames handed out in address order by ONE atomic per workgroup and tile
    // (requested one tile ahead), one workgroup barrier per tile
    __shared__ unsigned s_tile[2];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long ntiles = (nruns + wpb - 1) / wpb;
    if (threadIdx.x == 0) s_tile[0] = atomicAdd(counter, 1u);
    __syncthreads();
    for (int it = 0;; ++it) {
      const long long tile = s_tile[it & 1];
      if (tile >= ntiles) break;
      unsigned nxt = 0;
      if (threadIdx.x == 0) nxt = atomicAdd(counter, 1u);
      const long long r = tile * wpb + wv;
      if (r < nruns) {
        const long long f0 = r * G;
        long long f1 = f0 + G; if (f1 > total) f1 = total;
        if (KIND == 2) inv_run<NT>(spec_in, audio, f0, f1, lane);
        else fwd_run<NT, KIND == 1>(x, out, feat, f0, f1, lane);
      }
      if (threadIdx.x == 0) s_tile[(it + 1) & 1] = nxt;
      __syncthreads();
    }
  } else if (SHAPE == 6) {
    // persistent waves, each XCD walks its own eighth of the stream; a wave takes the next run of its XCD's eighth from
    // that XCD's counter (8 counters, 256 bytes apart), requested one run ahead
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID bits 0..3
    unsigned* ctr = counter + 64 * xcc;
    const long long per = (nruns + 7) / 8;
    const long long lo = per * xcc;
    long long hi = lo + per; if (hi > nruns) hi = nruns;
    unsigned nxt = 0;
    if (lane == 0) nxt = atomicAdd(ctr, 1u);
    long long r = lo + __builtin_amdgcn_readfirstlane(nxt);
    while (r < hi) {
      if (lane == 0) nxt = atomicAdd(ctr, 1u);
      const long long f0 = r * G;
      long long f1 = f0 + G; if (f1 > total) f1 = total;
      if (KIND == 2) inv_run<NT>(spec_in, audio, f0, f1, lane);
      else fwd_run<NT, KIND == 1>(x, out, feat, f0, f1, lane);
      r = lo + __builtin_amdgcn_readfirstlane(nxt);
    }
  } else if (SHAPE == 4) {
    // hybrid: workgroups in dispatch order, each owning a tile of wpb x K sub-runs of G frames; wave w of the workgroup
    // takes sub-runs w, w + wpb, w + 2 wpb, ... of the tile (nwaves carries K here)
    const long long K = nwaves;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long r0 = (long long)blockIdx.x * wpb * K;
    for (long long k = 0; k < K; ++k) {
      const long long r = r0 + k * wpb + wv;
      if (r >= nruns) break;
      const long long f0 = r * G;
      long long f1 = f0 + G; if (f1 > total) f1 = total;
      if (KIND == 2) inv_run<NT>(spec_in, audio, f0, f1, lane);
      else fwd_run<NT, KIND == 1>(x, out, feat, f0, f1, lane);
    }
  } else if (SHAPE == 0) {
    const long long w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
    if (lds_fill < 0) {     // de-phased start: waves begin up to ~64 x 64 clocks apart
      const unsigned h = ((unsigned)w * 2654435761u) >> 26;
      for (unsigned i = 0; i < h; ++i) __builtin_amdgcn_s_sleep(64);
    }
    for (long long r = w; r < nruns; r += nwaves) {
      const long long f0 = r * G;
      long long f1 = f0 + G; if (f1 > total) f1 = total;
      if (KIND == 2) inv_run<NT>(spec_in, audio, f0, f1, lane);
      else fwd_run<NT, KIND == 1>(x, out, feat, f0, f1, lane);
    }
  } else {
    unsigned nxt = 0;
    if (lane == 0) nxt = atomicAdd(counter, 1u);
    long long r = __builtin_amdgcn_readfirstlane(nxt);
    while (r < nruns) {
      if (lane == 0) nxt = atomicAdd(counter, 1u);      // requested now, consumed after this run
      const long long f0 = r * G;
      long long f1 = f0 + G; if (f1 > total) f1 = total;
      if (KIND == 2) inv_run<NT>(spec_in, audio, f0, f1, lane);
      else fwd_run<NT, KIND == 1>(x, out, feat, f0, f1, lane);
      r = __builtin_amdgcn_readfirstlane(nxt);
    }
  }
}

static unsigned* counter; static float* tables;
template <int KIND, int SHAPE, int NT>
static void run_pat(const char* name, long long G, int wpb, int bpc, int lds_bytes = 0, int lds_fill = 0) {
  const long long total = B * T;
  long long blocks, nwaves;
  if (SHAPE == 1 || SHAPE == 3) {
    const long long nruns = (total + G - 1) / G;
    blocks = (nruns + wpb - 1) / wpb; nwaves = blocks * wpb;
    if (SHAPE == 3) blocks = ((blocks + 7) / 8) * 8;
  } else if (SHAPE == 4) {
    const long long nruns = (total + G - 1) / G;
    const long long K = bpc;          // sub-runs per wave
    blocks = (nruns + wpb * K - 1) / (wpb * K); nwaves = K;
  } else {
    blocks = 256LL * bpc; nwaves = blocks * wpb;
    if (G <= 0) G = (total + nwaves - 1) / nwaves;
  }
  CHECK(hipFuncSetAttribute((const void*)pat_k<KIND, SHAPE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  float ms = time_ms([&] {
    if (SHAPE == 2 || SHAPE >= 5) CHECK(hipMemsetAsync(counter, 0, 4096, 0));
    hipLaunchKernelGGL((pat_k<KIND, SHAPE, NT>), dim3(blocks), dim3(64 * wpb), lds_bytes, 0, x, spec2, spec, audio, feat, total,
                       G, nwaves, lds_fill, tables, counter, sink);
  });
  const double bytes = KIND == 2 ? 5128.0 : KIND == 1 ? 5640.0 : 5128.0;
  static const char* shp[] = {"P ", "D ", "Pa", "Dx", "H ", "Pw", "Px"};
  printf("%-9s %s%s G=%-4lld wpb=%d lds=%3dK fill=%2dK %s=%-7lld %.3f ms  %.2f TB/s  frac %.3f\n", name, shp[SHAPE], NT ? " nt" : "   ",
         G, wpb, lds_bytes / 1024, lds_fill / 1024, (SHAPE & 1) ? "blocks" : "wavesCU", (SHAPE & 1) ? blocks : (long long)wpb * bpc, ms,
         total * bytes / ms / 1e9, total * bytes / ms / 8e9 / 1e3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t spec_bytes = B * T * 513 * 8 + 4096;
  CHECK(hipMalloc(&x, B * L * 4)); CHECK(hipMalloc(&spec, spec_bytes)); CHECK(hipMalloc(&spec2, spec_bytes));
  CHECK(hipMalloc(&audio, B * L * 4)); CHECK(hipMalloc(&feat, B * T * 128 * 4)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&counter, 4096)); CHECK(hipMalloc(&tables, 64 * 1024)); CHECK(hipMemset(tables, 0, 64 * 1024));
  CHECK(hipMemset(x, 0, B * L * 4)); CHECK(hipMemset(spec2, 0, spec_bytes));
  const char* what = argc > 1 ? argv[1] : "all";
  if (!strcmp(what, "all") || !strcmp(what, "fill")) fills();
  if (!strcmp(what, "all") || !strcmp(what, "pat")) {
    printf("== dispatch order (D): one workgroup per tile of wpb x G frames\n");
    for (int wpb : {1, 4, 8}) {
      for (long long G : {4LL, 8LL, 16LL, 32LL, 64LL}) {
        run_pat<0, 1, 1>("fwd", G, wpb, 0);
        run_pat<1, 1, 1>("fwd+feat", G, wpb, 0);
        run_pat<2, 1, 1>("inv", G, wpb, 0);
      }
    }
    run_pat<0, 1, 0>("fwd", 16, 4, 0);
    run_pat<2, 1, 0>("inv", 16, 4, 0);
    printf("== persistent (P): run r of trip i to wave (r mod nwaves)\n");
    for (int occ : {8, 16}) {
      for (long long G : {0LL, 4LL, 8LL, 16LL, 32LL}) {
        run_pat<0, 0, 1>("fwd", G, 8, occ / 8);
        run_pat<1, 0, 1>("fwd+feat", G, 8, occ / 8);
        run_pat<2, 0, 1>("inv", G, 8, occ / 8);
      }
    }
  }
  if (!strcmp(what, "hybrid")) {
    const int LDS = 64 * 1024, FILL = 22 * 1024;
    run_pat<1, 0, 1>("fwd+feat", 0, 8, 2, LDS, FILL);
    run_pat<1, 0, 1>("fwd+feat", 0, 8, 2, LDS, -1);        // de-phased
    run_pat<1, 0, 1>("fwd+feat", 8, 8, 2, LDS, FILL);
    run_pat<1, 0, 1>("fwd+feat", 8, 8, 2, LDS, -1);
    run_pat<0, 0, 1>("fwd", 0, 8, 2, LDS, FILL);
    run_pat<0, 0, 1>("fwd", 0, 8, 2, LDS, -1);
    for (long long G : {2LL, 4LL, 8LL}) {
      run_pat<1, 1, 1>("fwd+feat", G, 8, 0, LDS, FILL);
      run_pat<0, 1, 1>("fwd", G, 8, 0, LDS, FILL);
      for (int K : {2, 4, 8, 16, 32}) {
        run_pat<1, 4, 1>("fwd+feat", G, 8, K, LDS, FILL);
        run_pat<0, 4, 1>("fwd", G, 8, K, LDS, FILL);
      }
    }
  }
  if (!strcmp(what, "think")) {
    // the shapes again with a frame's worth of "compute" (s_sleep) between a frame's loads and its store burst
    const int LDS = 64 * 1024, FILL = 22 * 1024;
    for (int think : {0, 40, 80, 104, 120}) {
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_think), &think, sizeof(int)));
      printf("-- think = %d x 64 clocks per frame\n", think);
      run_pat<1, 0, 1>("fwd+feat", 0, 8, 2, LDS, FILL);
      for (long long G : {4LL, 8LL, 16LL}) {
        run_pat<1, 1, 1>("fwd+feat", G, 8, 0, LDS, FILL);
        run_pat<1, 5, 1>("fwd+feat", G, 8, 2, LDS, FILL);
      }
    }
  }
  if (!strcmp(what, "wide")) {
    const int LDS = 64 * 1024, FILL = 22 * 1024;
    for (int think : {0, 80, 104}) {
      CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_think), &think, sizeof(int)));
      for (int wide : {0, 1}) {
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_wide), &wide, sizeof(int)));
        printf("-- think = %d x 64 clocks per frame, %s stores\n", think, wide ? "16-B" : "8-B");
        run_pat<1, 0, 1>("fwd+feat", 0, 8, 2, LDS, FILL);
        run_pat<0, 0, 1>("fwd", 0, 8, 2, LDS, FILL);
        run_pat<1, 5, 1>("fwd+feat", 8, 8, 2, LDS, FILL);
      }
    }
  }
  if (!strcmp(what, "dyn")) {
    const int LDS = 64 * 1024, FILL = 22 * 1024;
    for (long long G : {2LL, 4LL, 8LL, 16LL}) {
      run_pat<1, 1, 1>("fwd+feat", G, 8, 0, LDS, FILL);
      run_pat<1, 5, 1>("fwd+feat", G, 8, 2, LDS, FILL);
      run_pat<1, 6, 1>("fwd+feat", G, 8, 2, LDS, FILL);
      run_pat<0, 1, 1>("fwd", G, 8, 0, LDS, FILL);
      run_pat<0, 5, 1>("fwd", G, 8, 2, LDS, FILL);
      run_pat<0, 6, 1>("fwd", G, 8, 2, LDS, FILL);
      run_pat<2, 1, 1>("inv", G, 8, 0, LDS, FILL);
      run_pat<2, 5, 1>("inv", G, 8, 2, LDS, FILL);
      run_pat<2, 6, 1>("inv", G, 8, 2, LDS, FILL);
    }
  }
  if (!strcmp(what, "shape")) {
    // the product kernels' footprint: 8-wave workgroups, 2 per CU (LDS 64 KB each), 22 KB of tables per workgroup
    const int LDS = 64 * 1024, FILL = 22 * 1024;
    for (long long G : {4LL, 6LL, 8LL, 12LL, 16LL, 24LL, 32LL}) {
      printf("-- G=%lld\n", G);
      run_pat<1, 1, 1>("fwd+feat", G, 8, 0, LDS, 0);
      run_pat<1, 1, 1>("fwd+feat", G, 8, 0, LDS, FILL);
      run_pat<1, 3, 1>("fwd+feat", G, 8, 0, LDS, FILL);
      run_pat<1, 2, 1>("fwd+feat", G, 8, 2, LDS, FILL);
      run_pat<1, 0, 1>("fwd+feat", G, 8, 2, LDS, FILL);
      run_pat<1, 1, 1>("fwd+feat", G, 4, 0, LDS / 2, FILL);
      run_pat<0, 1, 1>("fwd", G, 8, 0, LDS, FILL);
      run_pat<0, 2, 1>("fwd", G, 8, 2, LDS, FILL);
      run_pat<2, 1, 1>("inv", G, 8, 0, LDS, FILL);
      run_pat<2, 2, 1>("inv", G, 8, 2, LDS, FILL);
      run_pat<2, 0, 1>("inv", G, 8, 2, LDS, FILL);
      run_pat<2, 1, 1>("inv", G, 8, 0, 40 * 1024, FILL);   // 3 workgroups per CU (the inverse's occupancy)
    }
  }
  return 0;
}
