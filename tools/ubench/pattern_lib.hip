// pattern_lib.hip -- measurement helpers as a shared library (NOT product code; nothing under acids_transforms_amd/
// loads it).  bench.py and tools/step_probe.py call it through ctypes, on torch's stream, in the same process as the
// product kernels:
//   pat_fwd / pat_inv   copy kernels with the n_fft-1024 forward / inverse ACCESS PATTERN and no arithmetic -- the
//                       same kernels as tools/ubench/stream_pattern3.hip, shape "D" (one workgroup per tile of
//                       wpb x G frames, tiles dispatched in address order, non-temporal stores): what the memory
//                       system gives these exact streams on this box today;
//   ek_run              instruction-energy microbenchmarks: every wave of a chip-filling grid runs `iters` trips of an
//                       unrolled block of ONE kind of instruction; with the socket power sampled next to it that
//                       gives Joules per wave-instruction (profiles/r05_power_clock.md).
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libpattern.so tools/ubench/pattern_lib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float vf2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int F = 513;

typedef float vf4 __attribute__((ext_vector_type(4)));
template <int FEAT, int NT, int WIDE>
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
    // the row as whole 512-byte aligned blocks of the contiguous (frames, 513) complex stream
    if (WIDE) {      // 1-KB blocks, 16 bytes per lane
      const long long b0 = (f * F) >> 7, b1 = ((f + 1) * F) >> 7;
      int m = 0;
      for (long long blk = b0; blk < b1; ++blk, ++m) {
        const vf4 q = {v[(2 * m) & 7].x, v[(2 * m) & 7].y, v[(2 * m + 1) & 7].x, v[(2 * m + 1) & 7].y};
        vf4* dst = reinterpret_cast<vf4*>(out) + blk * 64 + lane;
        if (NT) __builtin_nontemporal_store(q, dst); else *dst = q;
      }
    } else {
      const long long b0 = (f * F) >> 6, b1 = ((f + 1) * F) >> 6;
      int m = 0;
      for (long long blk = b0; blk < b1; ++blk, ++m) {
        if (NT) __builtin_nontemporal_store(v[m & 7], base + blk * 64 + lane); else base[blk * 64 + lane] = v[m & 7];
      }
    }
    if (FEAT) {
      vf2* fd = reinterpret_cast<vf2*>(feat + f * 128);
      fd[lane] = v[0] + v[1];
    }
  }
}

template <int NT, int NTLOAD>
__device__ __forceinline__ void inv_run(const float2* __restrict__ in, float* __restrict__ audio, long long f0, long long f1,
                                        int lane) {
  vf2 acc[6];
#pragma unroll
  for (int m = 0; m < 6; ++m) acc[m] = (vf2){0.f, 0.f};
  for (long long f = f0; f < f1; ++f) {
    const vf2* src = reinterpret_cast<const vf2*>(in + f * F);
    vf2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = NTLOAD ? __builtin_nontemporal_load(&src[lane + 64 * m]) : src[lane + 64 * m];
    const float ny = reinterpret_cast<const float*>(src + 512)[0];
    vf2 o0 = acc[0] + v[0] + (vf2){ny, ny}, o1 = acc[1] + v[1];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = acc[m + 2] + v[m + 2];
    acc[4] = v[6];
    acc[5] = v[7];
    vf2* dst = reinterpret_cast<vf2*>(audio + f * 256);
    if (NT) {
      __builtin_nontemporal_store(o0, &dst[lane]);
      __builtin_nontemporal_store(o1, &dst[lane + 64]);
    } else {
      dst[lane] = o0;
      dst[lane + 64] = o1;
    }
  }
}

// forward + the reference's default 513 features per frame (2052-byte rows): STYLE 1 = every row written where it lies
// (4-byte stores, lane l -> floats l + 64 j: what the generic epilogue's scatter amounts to at best), STYLE 2 = the feature
// stream as whole 1-KB aligned blocks of 16 bytes per lane (what the packed epilogue writes)
template <int STYLE>
__global__ __launch_bounds__(512) void pat513_k(const float* __restrict__ x, float2* __restrict__ out, float* __restrict__ feat,
                                                long long total, long long G) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long long nruns = (total + G - 1) / G;
  const long long w = (long long)blockIdx.x * wpb + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (w >= nruns) return;
  const long long f0 = w * G;
  long long f1 = f0 + G;
  if (f1 > total) f1 = total;
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
    const long long b0 = (f * F) >> 6, b1 = ((f + 1) * F) >> 6;
    int m = 0;
    for (long long blk = b0; blk < b1; ++blk, ++m) __builtin_nontemporal_store(v[m & 7], base + blk * 64 + lane);
    if (STYLE == 1) {
      float* row = feat + f * F;
#pragma unroll
      for (int j = 0; j < 8; ++j) row[lane + 64 * j] = v[j].x + v[j].y;
      if (lane == 0) row[512] = v[0].x;
    } else {
      const long long q0 = (f * F) >> 8, q1 = ((f + 1) * F) >> 8;       // 256-float blocks
      int j = 0;
      for (long long blk = q0; blk < q1; ++blk, ++j) {
        const vf4 q = {v[(2 * j) & 7].x, v[(2 * j) & 7].y, v[(2 * j + 1) & 7].x, v[(2 * j + 1) & 7].y};
        *(reinterpret_cast<vf4*>(feat) + blk * 64 + lane) = q;
      }
    }
  }
}

// KIND 0 forward, 1 forward + 128 features, 2 inverse; NT: non-temporal stores; ALT: 16-byte stores (forward) / plain loads (inverse)
template <int KIND, int NT, int ALT>
__global__ __launch_bounds__(512) void pat_k(const float* __restrict__ x, const float2* __restrict__ spec_in,
                                             float2* __restrict__ out, float* __restrict__ audio, float* __restrict__ feat,
                                             long long total, long long G) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const long long nruns = (total + G - 1) / G;
  const long long w = (long long)blockIdx.x * wpb + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (w >= nruns) return;
  const long long f0 = w * G;
  long long f1 = f0 + G;
  if (f1 > total) f1 = total;
  if (KIND == 2) inv_run<NT, !ALT>(spec_in, audio, f0, f1, lane);
  else fwd_run<KIND == 1, NT, ALT>(x, out, feat, f0, f1, lane);
}

// ---------------------------------------------------------------------------------------------- instruction energy
// One kind of instruction, 64 per trip (8 independent chains of 8), so that issue is never dependency-bound at four
// waves per SIMD.
enum { EK_SLEEP = 0, EK_PK_FMA = 1, EK_FMA = 2, EK_MOV = 3, EK_CNDMASK = 4, EK_LDS_READ64 = 5, EK_LDS_WRITE64 = 6,
       EK_BPERMUTE = 7, EK_LDS_READ128 = 8, EK_SQRT = 9, EK_PK_ADD = 10, EK_SALU = 11, EK_COUNT = 12 };

template <int KIND>
__global__ __launch_bounds__(256) void ek_k(int iters, float* __restrict__ sink) {
  __shared__ float4 lds4[1024];
  const int lane = threadIdx.x & 63;
  vf2 a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (vf2){1.0f + 1e-3f * (lane + i), 0.5f - 1e-3f * i};
  const vf2 c = {0.999f, 1.0005f}, d = {1e-4f, -1e-4f};
  for (int i = threadIdx.x; i < 1024; i += 256) lds4[i] = make_float4(i, 1.f, 2.f, 3.f);
  __syncthreads();
  const vf2* lds2 = reinterpret_cast<const vf2*>(lds4) + (threadIdx.x & 255);
  vf2* ldsw = reinterpret_cast<vf2*>(lds4) + (threadIdx.x & 255);
  int idx = ((lane * 7) & 63) * 4;
  unsigned s = iters;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == EK_SLEEP) {
          if (i == 0) __builtin_amdgcn_s_sleep(16);
        } else if (KIND == EK_PK_FMA) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        } else if (KIND == EK_FMA) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(c.x), "v"(d.x));
        } else if (KIND == EK_MOV) {
          asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(a[(i + 1) & 7].y));
        } else if (KIND == EK_CNDMASK) {
          asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i].x) : "v"(a[(i + 1) & 7].y), "v"(c.x) : );
        } else if (KIND == EK_LDS_READ64) {
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[i]) : "v"((int)(uintptr_t)lds2 & 0xffff), "n"(0));
        } else if (KIND == EK_LDS_WRITE64) {
          asm volatile("ds_write_b64 %0, %1" : : "v"((int)(uintptr_t)ldsw & 0xffff), "v"(a[i]) : "memory");
        } else if (KIND == EK_BPERMUTE) {
          asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i].x) : "v"(idx));
        } else if (KIND == EK_LDS_READ128) {
          float4 q;
          asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"((int)((uintptr_t)(lds4 + (threadIdx.x & 255))) & 0xffff));
          a[i].x = q.x;
        } else if (KIND == EK_SQRT) {
          asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
        } else if (KIND == EK_PK_ADD) {
          asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(d));
        } else if (KIND == EK_SALU) {
          asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
        }
      }
    }
    if (KIND == EK_LDS_READ64 || KIND == EK_LDS_READ128 || KIND == EK_BPERMUTE || KIND == EK_LDS_WRITE64)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float acc = (float)s;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
  if (acc == 12345.678f) sink[0] = acc;
}

// flags: bit 0 = plain (temporal) stores instead of non-temporal; bit 1 = 16-byte stores (forward) / plain loads (inverse)
template <int KIND>
static void (*pick(int flags))(const float*, const float2*, float2*, float*, float*, long long, long long) {
  switch (flags & 3) {
    case 0: return pat_k<KIND, 1, 0>;
    case 1: return pat_k<KIND, 0, 0>;
    case 2: return pat_k<KIND, 1, 1>;
    default: return pat_k<KIND, 0, 1>;
  }
}

}  // namespace

extern "C" {
// flags: bit 0 = plain (temporal) stores instead of non-temporal; bit 1 = 16-byte stores (forward) / plain loads (inverse)

int pat_fwd_flags(const float* x, void* out, float* feat, long long total_frames, int G, int wpb, int flags, void* stream) {
  if (G < 1 || wpb < 1 || wpb > 8) return -1;
  const long long nruns = (total_frames + G - 1) / G;
  const long long blocks = (nruns + wpb - 1) / wpb;
  if (blocks <= 0 || blocks >= (1LL << 31)) return -1;
  auto k = feat ? pick<1>(flags) : pick<0>(flags);
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * wpb), 0, (hipStream_t)stream, x, (const float2*)nullptr, (float2*)out,
                     (float*)nullptr, feat, total_frames, (long long)G);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int pat_inv_flags(const void* spec, float* audio, long long total_frames, int G, int wpb, int flags, void* stream) {
  if (G < 1 || wpb < 1 || wpb > 8) return -1;
  const long long nruns = (total_frames + G - 1) / G;
  const long long blocks = (nruns + wpb - 1) / wpb;
  if (blocks <= 0 || blocks >= (1LL << 31)) return -1;
  hipLaunchKernelGGL(pick<2>(flags), dim3((unsigned)blocks), dim3(64 * wpb), 0, (hipStream_t)stream, (const float*)nullptr,
                     (const float2*)spec, (float2*)nullptr, audio, (float*)nullptr, total_frames, (long long)G);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

// forward + 513 features per frame; style 1 rows, 2 aligned 1-KB blocks
int pat_fwd513(const float* x, void* out, float* feat, long long total_frames, int G, int wpb, int style, void* stream) {
  if (G < 1 || wpb < 1 || wpb > 8 || !feat) return -1;
  const long long nruns = (total_frames + G - 1) / G;
  const long long blocks = (nruns + wpb - 1) / wpb;
  if (blocks <= 0 || blocks >= (1LL << 31)) return -1;
  if (style == 1)
    hipLaunchKernelGGL(pat513_k<1>, dim3((unsigned)blocks), dim3(64 * wpb), 0, (hipStream_t)stream, x, (float2*)out, feat, total_frames, (long long)G);
  else
    hipLaunchKernelGGL(pat513_k<2>, dim3((unsigned)blocks), dim3(64 * wpb), 0, (hipStream_t)stream, x, (float2*)out, feat, total_frames, (long long)G);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int pat_fwd(const float* x, void* out, float* feat, long long total_frames, int G, int wpb, void* stream) {
  return pat_fwd_flags(x, out, feat, total_frames, G, wpb, 0, stream);
}

int pat_inv(const void* spec, float* audio, long long total_frames, int G, int wpb, void* stream) {
  return pat_inv_flags(spec, audio, total_frames, G, wpb, 0, stream);
}

// `blocks` workgroups of four waves; every wave issues iters x 64 instructions of `kind` (EK_SLEEP: iters x 8 s_sleep 16)
int ek_run(int kind, int blocks, int iters, float* sink, void* stream) {
  void (*k)(int, float*) = nullptr;
  switch (kind) {
    case EK_SLEEP: k = ek_k<EK_SLEEP>; break;
    case EK_PK_FMA: k = ek_k<EK_PK_FMA>; break;
    case EK_FMA: k = ek_k<EK_FMA>; break;
    case EK_MOV: k = ek_k<EK_MOV>; break;
    case EK_CNDMASK: k = ek_k<EK_CNDMASK>; break;
    case EK_LDS_READ64: k = ek_k<EK_LDS_READ64>; break;
    case EK_LDS_WRITE64: k = ek_k<EK_LDS_WRITE64>; break;
    case EK_BPERMUTE: k = ek_k<EK_BPERMUTE>; break;
    case EK_LDS_READ128: k = ek_k<EK_LDS_READ128>; break;
    case EK_SQRT: k = ek_k<EK_SQRT>; break;
    case EK_PK_ADD: k = ek_k<EK_PK_ADD>; break;
    case EK_SALU: k = ek_k<EK_SALU>; break;
    default: return -1;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

int ek_kinds() { return EK_COUNT; }

}  // extern "C"
