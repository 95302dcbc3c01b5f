// pghi.hip -- phase-gradient heap integration (PGHI) on gfx950.
//
// Replaces the reference's pure-Python heap loops:
//   DGT.modgabphasegrad / pghi / perform_hgi              transforms/dgt.py:156-236   (K13, K14)
//   RealtimeDGT.modgabphasegrad / pghi / perform_hgi      transforms/dgt.py:338-466
//   utils/heapq.py:9-59 (binary min-heap on keys only, strict '<', right child on ties)
//
// The integration order is part of the contract: it is
// defined by exact fp32 compares of magnitudes and by the heap's tie-breaking,
// so the heap here is the same array-embedded binary heap with the same
// sift rules.  One wavefront owns one clip (offline) or one stream (realtime):
// the flood is inherently serial per clip, the batch supplies the parallelism.
// Wave-wide work (gradients, maxima / reseeds, thresholding) is lane-parallel.
//
// This file is compiled with -ffp-contract=off: every fp32 expression keeps
// the reference's rounding sequence (no FMA contraction).
#include <hip/hip_runtime.h>
#include "fastmath.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/acids_hip.h"
#include "variants.h"

namespace at_hip {

struct HeapItem {
  float key;  // -magnitude
  int idx;    // row * F + col
};

// ---- heap primitives, single lane (utils/heapq.py) --------------------------
__device__ __forceinline__ void h_siftdown(HeapItem* h, int startpos, int pos) {
  HeapItem newitem = h[pos];
  while (pos > startpos) {
    const int parentpos = (pos - 1) >> 1;
    const HeapItem parent = h[parentpos];
    if (newitem.key < parent.key) {
      h[pos] = parent;
      pos = parentpos;
      continue;
    }
    break;
  }
  h[pos] = newitem;
}

__device__ __forceinline__ void h_siftup(HeapItem* h, int endpos, int pos) {
  const int startpos = pos;
  const HeapItem newitem = h[pos];
  int childpos = 2 * pos + 1;
  while (childpos < endpos) {
    const int rightpos = childpos + 1;
    if (rightpos < endpos && !(h[childpos].key < h[rightpos].key)) childpos = rightpos;
    h[pos] = h[childpos];
    pos = childpos;
    childpos = 2 * pos + 1;
  }
  h[pos] = newitem;
  h_siftdown(h, startpos, pos);
}

__device__ __forceinline__ void h_push(HeapItem* h, int& n, float key, int idx) {
  h[n].key = key;
  h[n].idx = idx;
  ++n;
  h_siftdown(h, 0, n - 1);
}

__device__ __forceinline__ HeapItem h_pop(HeapItem* h, int& n) {
  const HeapItem last = h[n - 1];
  --n;
  if (n > 0) {
    const HeapItem ret = h[0];
    h[0] = last;
    h_siftup(h, n, 0);
    return ret;
  }
  return last;
}

// ---- wave-wide (value, first index) arg-max --------------------------------
__device__ __forceinline__ void wave_argmax(float& v, long long& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const long long oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) {
      v = ov;
      i = oi;
    }
  }
}

// ---------------------------------------------------------------------------
// K13 offline: s = clamp(mag, eps); log; replicate-padded central differences
// ---------------------------------------------------------------------------
struct GradParams {
  const float* mag;  // (B, T, F)
  float* spec;       // (B, T, F) clamped work copy (may be null)
  float* tgradw;
  float* fgradw;
  long long B;
  int T, F, n_fft, hop;
  float gamma, eps;
};

__global__ __launch_bounds__(256) void pghi_grad_offline_kernel(GradParams p) {
  const float fmul = p.gamma / (float)((long long)p.hop * (long long)p.n_fft);
  const float fstep = ((float)(2.0 * 3.14159265358979323846) * (float)p.hop) / (float)p.n_fft;
  const float pi_f = (float)3.14159265358979323846;
  const long long per = (long long)p.T * p.F;
  const long long total = p.B * per;
  // (clip, frame, bin) of the flat index are carried along the grid stride: two 64-bit divisions per thread instead of
  // two per element (1.63 -> 1.43 ms per 1024 clips, tools/grad_probe.py; the rest is four logf per bin)
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long b = i0 / per;
  const long long r0 = i0 - b * per;
  int t = (int)(r0 / p.F), k = (int)(r0 - (long long)t * p.F);
  const long long sb = stride / per;
  const long long sr = stride - sb * per;
  const int st = (int)(sr / p.F), sk = (int)(sr - (long long)st * p.F);
  for (long long i = i0; i < total; i += stride, b += sb, t += st, k += sk) {
    if (k >= p.F) {
      k -= p.F;
      ++t;
    }
    if (t >= p.T) {
      t -= p.T;
      ++b;
    }
    const long long r = (long long)t * p.F + k;
    const float* m = p.mag + b * per;
    const int tu = t + 1 < p.T ? t + 1 : p.T - 1, td = t > 0 ? t - 1 : 0;
    const int kr = k + 1 < p.F ? k + 1 : p.F - 1, kl = k > 0 ? k - 1 : 0;
    const float c = fmaxf(m[r], p.eps);
    const float right = logf(fmaxf(m[(long long)t * p.F + kr], p.eps));
    const float left = logf(fmaxf(m[(long long)t * p.F + kl], p.eps));
    const float up = logf(fmaxf(m[(long long)tu * p.F + k], p.eps));
    const float dn = logf(fmaxf(m[(long long)td * p.F + k], p.eps));
    const float dxdw = (right - left) / 2.0f;
    const float dxdt = (up - dn) / 2.0f;
    p.fgradw[i] = dxdw / fmul + fstep * (float)k;
    p.tgradw[i] = (-fmul) * dxdt + pi_f;
    if (p.spec) p.spec[i] = c;
  }
}

// ---------------------------------------------------------------------------
// K14 offline: one wave per clip
// ---------------------------------------------------------------------------
struct HgiParams {
  float* spec;          // (B, T, F) clamped magnitudes, consumed (visited cells <- abstol)
  const float* tgradw;  // (B, T, F)
  const float* fgradw;
  float* phase;         // (B, T, F) output
  HeapItem* heap;       // (B, T*F + 2)
  long long B;
  int T, F;
  float abstol, tol;
  long long* npops;     // optional (B) number of pops per clip
  int heap_lds_cap;     // heap entries kept in LDS per clip (2^k - 1)
  int seg_cap;          // cooperative kernel: segment maxima kept in LDS per clip (reseeding), behind the heap's top
  int prof;             // dev only (ACIDS_PGHI_PROF=1): cycle counters go to `order` instead of the pop order
  int* order;           // optional (B, T*F) pop order (row*F+col), for the parity tests
};

// The reference rewrites every cell below max*tol to abstol up front
// (dgt.py:177-178); here that threshold is applied when a cell is read.
__device__ __forceinline__ bool live(float v, float abstol, float thr) { return v > abstol && !(v < thr); }

// global (value, first row-major index) maximum over the live cells of one clip.  Eight independent loads per trip:
// a plain one-load loop pays a full memory round trip per 64 cells (2.5 ms per scan of a 4-second clip).
__device__ __forceinline__ void clip_argmax(const float* spec, long long n, float abstol, float thr, bool use_thr,
                                            int lane, float& best, long long& besti) {
  float v = -1.0f;
  long long vi = n;
  for (long long base = 0; base < n; base += 512) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = base + lane + 64 * u;
      x[u] = spec[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = base + lane + 64 * u;
      float sv = x[u];
      if (use_thr && sv < thr) sv = abstol;
      if (i < n && sv > v) {
        v = sv;
        vi = i;
      }
    }
  }
  wave_argmax(v, vi);
  best = v;
  besti = vi;
}

// ---- reseeding without rescanning the clip -------------------------------------------------------------------
// dgt.py:216-219 takes the global maximum of what is left every time the heap runs empty; a decaying sound does that
// hundreds of times per clip (301 seeds in one second of decaying noise), and a full scan of a
// 4-second clip is 354 k cells.  The clip is cut into S <= seg_cap segments of SL cells (row-major order) with an
// UPPER BOUND of each segment's live maximum in LDS: exact at the start, stale-high afterwards (the flood only
// lowers cells).  A reseed takes the first segment holding the largest bound, rescans that one segment, and is done
// if the bound was exact -- every earlier segment has a smaller bound, every later one at most the same -- otherwise
// it repairs the bound and repeats.  The result is the full scan's (value, first row-major index), cell for cell.
struct SegMax {
  float* m;        // LDS, S entries
  int S;
  long long SL;    // cells per segment (multiple of 512)
};

// maximum of a non-negative value over the wave (DPP row reduction + row broadcasts), valid on every lane
__device__ __forceinline__ float wave_max_nonneg(float v) {
  int x = (int)__float_as_uint(v);
  auto mx = [](int a, int b) { return (int)__float_as_uint(fmaxf(__uint_as_float((unsigned)a), __uint_as_float((unsigned)b))); };
  x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true));   // row_shr:1
  x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true));   // row_shr:2
  x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true));   // row_shr:4
  x = mx(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true));   // row_shr:8  -> lane 15 of each row
  x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1, 3
  x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2, 3 -> lane 63
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane(x, 63));
}

// one segment: (value, first index) maximum under the threshold rule (use_thr) -- at most SL / 64 cells per lane
__device__ __forceinline__ void seg_scan(const float* spec, long long lo, long long hi, float abstol, float thr, bool use_thr,
                                         int lane, float& v, long long& vi) {
  for (long long base = lo; base < hi; base += 512) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = base + lane + 64 * u;
      x[u] = spec[i < hi ? i : hi - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = base + lane + 64 * u;
      float sv = x[u];
      if (use_thr && sv < thr) sv = abstol;
      if (i < hi && sv > v) {
        v = sv;
        vi = i;
      }
    }
  }
}

// exact bounds for every segment (+ the clip's global maximum): the first scan (use_thr = false), and the repair of
// last resort when a reseed keeps hitting stale bounds
__device__ __forceinline__ void seg_rebuild(const float* spec, long long n, const SegMax& G, float abstol, float thr,
                                            bool use_thr, int lane, float& best, long long& besti) {
  float bv = -1.0f;
  long long bi = n;
  for (int sg = 0; sg < G.S; ++sg) {
    const long long lo = sg * G.SL, hi = (lo + G.SL < n) ? lo + G.SL : n;
    float v = -1.0f;
    long long vi = n;
    seg_scan(spec, lo, hi, abstol, thr, use_thr, lane, v, vi);
    const float sm = wave_max_nonneg(fmaxf(v, 0.0f));
    if (lane == 0) G.m[sg] = sm;
    if (v > bv) {      // per lane: cells are visited in increasing index order, so `>` keeps the first index
      bv = v;
      bi = vi;
    }
  }
  wave_argmax(bv, bi);
  best = bv;
  besti = bi;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

__device__ __forceinline__ void seg_reseed(const float* spec, long long n, const SegMax& G, float abstol, float thr,
                                           int lane, float& max_val, long long& max_pos) {
  for (int tries = 0;; ++tries) {
    if (tries == 48) {      // many stale bounds in a row (e.g. the one reseed at the end of a dense clip): rebuild them all
      seg_rebuild(spec, n, G, abstol, thr, true, lane, max_val, max_pos);
      return;
    }
    float bv = -1.0f;
    long long bs = G.S;
    for (int sg = lane; sg < G.S; sg += 64) {
      const float x = G.m[sg];
      if (x > bv) {
        bv = x;
        bs = sg;
      }
    }
    wave_argmax(bv, bs);
    if (!(bv > abstol)) {   // nothing live anywhere: the caller's loop ends (any in-range position will do)
      max_val = abstol;
      max_pos = 0;
      return;
    }
    const long long lo = bs * G.SL, hi = (lo + G.SL < n) ? lo + G.SL : n;
    float tv = -1.0f;
    long long ti = n;
    seg_scan(spec, lo, hi, abstol, thr, true, lane, tv, ti);
    wave_argmax(tv, ti);
    if (tv == bv) {
      max_val = tv;
      max_pos = ti;
      return;
    }
    if (lane == 0) G.m[bs] = tv;   // stale-high bound repaired; try again
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

__global__ __launch_bounds__(64) void pghi_hgi_offline_kernel(HgiParams p) {
  const long long b = blockIdx.x;
  if (b >= p.B) return;
  const int lane = threadIdx.x;
  const int T = p.T, F = p.F;
  const long long n = (long long)T * F;
  float* spec = p.spec + b * n;
  const float* tg = p.tgradw + b * n;
  const float* fg = p.fgradw + b * n;
  float* phase = p.phase + b * n;
  HeapItem* heap = p.heap + b * (n + 2);
  int* order = p.order ? p.order + b * n : nullptr;
  const float abstol = p.abstol;

  for (long long i = lane; i < n; i += 64) phase[i] = 0.0f;  // dgt.py:170

  float max_val;
  long long max_pos;
  clip_argmax(spec, n, abstol, 0.f, false, lane, max_val, max_pos);  // :173-174
  const float thr = max_val * p.tol;                                   // :177-178
  long long npops = 0;
  int hn = 0;
  if (lane == 0) {
    heap[0].key = -max_val;  // :175
    heap[0].idx = (int)max_pos;
    spec[max_pos] = abstol;  // :176
  }
  hn = 1;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  while (max_val > abstol) {  // :179
    if (lane == 0) {
      while (hn > 0) {  // :180
        const HeapItem it = h_pop(heap, hn);
        if (order) order[npops] = it.idx;
        ++npops;
        const int c = it.idx;
        const int col = c / F;      // frame
        const int row = c - col * F;  // bin
        const float pc = phase[c];
        if (col < T - 1) {  // :188-194
          const float s = spec[c + F];
          if (live(s, abstol, thr)) {
            phase[c + F] = pc + (fg[c] + fg[c + F]) / 2.0f;
            h_push(heap, hn, -s, c + F);
            spec[c + F] = abstol;
          }
        }
        if (col > 0) {  // :195-201
          const float s = spec[c - F];
          if (live(s, abstol, thr)) {
            phase[c - F] = pc - (fg[c] + fg[c - F]) / 2.0f;
            h_push(heap, hn, -s, c - F);
            spec[c - F] = abstol;
          }
        }
        if (row < F - 1) {  // :202-208
          const float s = spec[c + 1];
          if (live(s, abstol, thr)) {
            phase[c + 1] = pc + (tg[c] + tg[c + 1]) / 2.0f;
            h_push(heap, hn, -s, c + 1);
            spec[c + 1] = abstol;
          }
        }
        if (row > 0) {  // :209-215
          const float s = spec[c - 1];
          if (live(s, abstol, thr)) {
            phase[c - 1] = pc - (tg[c] + tg[c - 1]) / 2.0f;
            h_push(heap, hn, -s, c - 1);
            spec[c - 1] = abstol;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __threadfence_block();
    // :216-219 reseed from the global max of what is left (lane-parallel scan)
    clip_argmax(spec, n, abstol, thr, true, lane, max_val, max_pos);
    if (lane == 0) {
      heap[0].key = -max_val;
      heap[0].idx = (int)max_pos;
      spec[max_pos] = abstol;
    }
    hn = 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  if (p.npops && lane == 0) p.npops[b] = npops;
}

// ---------------------------------------------------------------------------
// K14 offline, wave-cooperative heap.  Same binary heap, same sift rules, same
// pop order as the single-lane version above -- but every heap operation is done
// by the whole wavefront so that its ~log2(n) *dependent* memory accesses become
// a few wide ones:
//   pop  : the bubble-up path from the hole is resolved five levels per round:
//          63 lanes gather the depth-6 subtree under the hole (node i of the
//          subtree on lane i), every inner lane picks its smaller child (right
//          child on ties, utils/heapq.py:33), the path is read off with five
//          v_readlane steps and all moved entries are written by one store;
//   push / final sift-down: every ancestor of the insertion point is known from
//          its index alone, so lane L loads ancestor L, one ballot finds how far
//          the item rises (strict '<', heapq.py:16) and one store shifts the chain;
//   neighbours: lanes 0-3 handle next-frame / prev-frame / next-bin / prev-bin.
// Heap words are read with agent-scope loads (L2-served): entries written by one
// lane are re-read by other lanes of the same wave a few instructions later.
// ---------------------------------------------------------------------------
typedef unsigned long long u64;

__device__ __forceinline__ u64 pack_item(float key, int idx) {
  return ((u64)__float_as_uint(key) << 32) | (unsigned)idx;
}
__device__ __forceinline__ float item_key(u64 e) { return __uint_as_float((unsigned)(e >> 32)); }
__device__ __forceinline__ int item_idx(u64 e) { return (int)(unsigned)e; }

// Heap words and cell state in global memory are written and re-read by lanes of ONE wave only.  A CU's vector L1
// is write-through and coherent for the waves of that CU, so workgroup scope is all the visibility this needs
// (ACIDS_PGHI_SCOPE=__HIP_MEMORY_SCOPE_AGENT at compile time restores the L2-served sc1 accesses: those write
// through to memory and DROP the line from L2 -- MI355X_MICROARCH.md, "stores of each flavour" -- so every later read
// of a heap entry paid a trip beyond L2).  The relaxed atomics only pin the compiler's ordering.
#ifndef ACIDS_PGHI_SCOPE
#define ACIDS_PGHI_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
__device__ __forceinline__ u64 gload(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, ACIDS_PGHI_SCOPE);
}
__device__ __forceinline__ void gstore(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, ACIDS_PGHI_SCOPE);
}

// The heap array: positions [0, cap) live in this wave's LDS (the top levels, where every pop starts),
// the rest in the clip's global workspace.  Same array, same indices -- only the storage differs.
// cap = 2^k - 1 is chosen at launch from the batch size: 4095 entries (32 KB) while <= 4 clips share a
// CU, fewer when more clips have to be resident at once.
// SPILLS = false: the whole heap is in LDS (realtime kernel) and the global paths compile away.
template <bool SPILLS>
struct HeapT {
  u64* top;   // LDS, cap entries
  u64* rest;  // global, indexed by absolute position
  int cap;
  __device__ __forceinline__ u64 load(long long pos) const {
    if constexpr (!SPILLS) return top[pos];
    return pos < cap ? top[pos] : gload(rest + pos);
  }
  __device__ __forceinline__ void store(long long pos, u64 v) const {
    if constexpr (!SPILLS) {
      top[pos] = v;
      return;
    }
    if (pos < cap) top[pos] = v;
    else gstore(rest + pos, v);
  }
  // entry `pos` for the lanes that `want` it, `dflt` for the others.  The LDS read is unconditional (slot 0 for
  // lanes that do not want it or whose entry is global): one divergent region -- the global load -- instead of
  // a nest of three, and none at all while the whole subtree is in LDS.
  __device__ __forceinline__ u64 load_if(int pos, bool want, u64 dflt) const {
    if constexpr (!SPILLS) {
      const u64 v = top[want ? pos : 0];
      return want ? v : dflt;
    }
    const bool in_lds = pos < cap;
    u64 v = top[(want && in_lds) ? pos : 0];
    if (want && !in_lds) {
      v = gload(rest + pos);
      // Wait for it here, inside the branch.  Left to the compiler, the wait lands after the join as vmcnt(0),
      // and rounds that touched LDS only would then sit out the neighbourhood loads the pop has in flight
      // (memory returns in order): an HBM round trip exposed on every pop.
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    }
    return want ? v : dflt;
  }
};
typedef HeapT<true> Heap;
__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  const unsigned lo = __shfl((unsigned)v, src, 64);
  const unsigned hi = __shfl((unsigned)(v >> 32), src, 64);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float fload(const float* p) {   // a cell another lane of this wave may have just written
  return __hip_atomic_load(p, __ATOMIC_RELAXED, ACIDS_PGHI_SCOPE);
}

// place `item` at `pos` and let it rise (utils/heapq.py:9-21 with startpos = 0)
template <typename HEAP>
__device__ __forceinline__ void coop_siftdown(const HEAP& H, int pos, u64 item, int lane) {
  const unsigned q = (unsigned)pos + 1u;
  const int depth = 31 - __clz(q);                // number of ancestors (< 31)
  const int sh = lane < 31 ? lane : 30;           // lanes >= depth are idle; keep their shifts defined
  const int my_dst = (int)(q >> sh) - 1;          // lane L: position of ancestor L-1 (L = 0: pos itself)
  const int my_anc = (int)(q >> (sh + 1)) - 1;    // lane L: position of ancestor L
  const u64 anc = H.load_if(my_anc, lane < depth, 0);
  const bool rises = (lane < depth) && (item_key(item) < item_key(anc));
  const u64 mask = __ballot(rises);
  const int m = (mask == ~0ull) ? 64 : __builtin_ctzll(~mask);  // item passes ancestors 0 .. m-1
  // lanes 0 .. m-1 move their ancestor one step down, lane m drops the item: one store site
  if (lane <= m) H.store(my_dst, lane == m ? item : anc);
}

// sibling's value through DPP (lane ^ 1): pure VALU, no LDS crossbar
__device__ __forceinline__ float dpp_xor1(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ int dpp_xor1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }

__device__ __forceinline__ u64 readlane64(u64 v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

// utils/heapq.py:51-59 (+ :24-42): the bubble-up part of heappop after `last` was taken off the end
// (n = remaining size >= 1).  Returns the leaf position where `last` has to be placed.
// lanes {L, L >> 1, L >> 2, ...} >= 2: the nodes that must all be chosen children for local node L to bubble up
__device__ __forceinline__ u64 chain_mask(int lane) {
  u64 m = 0;
  for (int a = lane; a >= 2; a >>= 1) m |= 1ull << a;
  return m;
}

// `top63`: positions 0..62 as lane L - 1 holds them (lane 0: anything), read by the caller before the pop began --
// the first round always gathers from there, so its load is off the critical path.
template <typename HEAP>
__device__ __forceinline__ int coop_bubble(const HEAP& H, int n, int lane, u64 anc_mask, u64& leaf_old, u64 top63) {
  int pos = 0;  // the hole
  const int lvl = 31 - __clz((unsigned)lane | 1u);
  const int off = lane - (1 << lvl);
  const u64 kInf = (u64)0x7f800000u << 32;
  // Rounds are aligned to the *bottom* of the heap: the first one descends only ((D - 1) mod 5) + 1 levels
  // (D = the last level), so that the last round covers levels D-4 .. D.  With the top 12 levels in LDS a heap
  // of up to 2^17 entries then pays one global round per pop, where top-aligned rounds (1-5, 6-10, 11-15, 16)
  // pay two as soon as D = 16 -- a third of all pops on dense spectra.  Same number of rounds either way.
  const int last_level = 31 - __clz((unsigned)n | 1u);
  int limit = last_level >= 1 ? ((last_level - 1) % 5) + 1 : 5;
  bool first = true;
#ifndef AT_PGHI_NO_ROOT_STEP
  if (limit == 1) {
    // A first round of ONE level (last level 1, 6, 11 or 16 -- a third of all pops on dense spectra sit at 16) is the
    // root choosing between its two children: both are in `top63` (positions 1 and 2 on lanes 2 and 3), so the round is
    // two readlanes, one compare (heapq.py:33: the right child unless left < right; a missing child reads +inf) and one
    // store, all on the scalar side, instead of the 64-lane machinery below and its LDS round trip.
    const u64 v1 = readlane64(top63, 2);
    const u64 v2 = (2 < n) ? readlane64(top63, 3) : kInf;
    const bool left = item_key(v1) < item_key(v2);
    const u64 vc = left ? v1 : v2;
    pos = left ? 1 : 2;
    if (lane == 0) H.store(0, vc);
    if (2 * pos + 1 >= n) {
      leaf_old = vc;
      return pos;
    }
    limit = 5;
    first = false;
  }
#endif
  for (;;) {
    // subtree under the hole: local node `lane` (1..63) <-> global index g
    const int g = ((pos + 1) << lvl) - 1 + off;           // < 2^25: heap positions are < T F < 2^31 >> 5
    const bool valid = (lane >= 1) && (lvl <= limit) && (g < n);
    const u64 val = first ? (valid ? top63 : kInf) : H.load_if(g, valid, kInf);
    first = false;
    const float key = item_key(val);
    // "am I the child my parent bubbles up?"  children 2i (left, even lane) and 2i+1 (right, odd lane) are
    // DPP neighbours.  heapq.py:33: take the right child iff it exists and not (left < right); a missing child
    // reads as +inf (real keys are -magnitude, finite), so one compare `left < right` decides for both lanes.
    // As lane masks (scalar unit): left children are chosen where key < sibling, right ones where not
    // (sibling < key).
    const float sib = dpp_xor1(key);
    const u64 kOdd = 0xAAAAAAAAAAAAAAAAull;
    const u64 m_lt = __ballot(key < sib), m_gt = __ballot(sib < key);
    const u64 W = __ballot(valid) & ((~kOdd & m_lt) | (kOdd & ~m_gt)) & ~3ull;
    // The chain of bubbled-up nodes below the subtree root: node L belongs to it iff L and every ancestor of
    // L down to level 1 is its parent's chosen child, i.e. iff W covers the lane's ancestor mask; below a
    // leaf no bit is set, so the chain simply ends there.
    const bool on_chain = (lane >= 2) && ((W & anc_mask) == anc_mask);
    const u64 chain = __ballot(on_chain);
    const int steps = __builtin_popcountll(chain);
    const int cur = steps ? 63 - __builtin_clzll(chain) : 1;          // the final hole of this round
    // every bubbled-up entry moves into its parent's slot (the hole, or the chain node above it)
    if (on_chain) H.store((g - 1) >> 1, val);
    const int gcur = __builtin_amdgcn_readlane(g, cur);
    pos = gcur;
    if (steps < limit || 2 * gcur + 1 >= n) {
      leaf_old = readlane64(val, cur);   // what the final hole held: now the value of its parent
      break;
    }
    limit = 5;
  }
  return pos;
}

// PUSH_BATCH (opt-in, ACIDS_PGHI_PUSH=batch): round-3 experiment, measured slower -- see the comment at the pushes.
template <bool PROF, bool PUSH_BATCH = false>
__global__ __launch_bounds__(512) void pghi_hgi_offline_coop_kernel(HgiParams p) {
  // one wave per clip; 1, 2, 4 or 8 waves per workgroup (independent: no workgroup-level synchronisation).  A
  // workgroup's waves are spread evenly over the CU's four SIMDs, whereas 64-thread workgroups are placed by the
  // dispatcher as it sees fit -- and a SIMD that is handed one clip more than its neighbours finishes them all later:
  // the kernel's tail (4096 clips: 2.00 s as 4096 single-wave workgroups, 1.53 s as 512 eight-wave ones).
  // The wave number is uniform by construction; said explicitly (readfirstlane), the clip's base pointers, the heap
  // descriptor and the segment table live in scalar registers and every cell / heap address is base + 32-bit offset.
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long b = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (b >= p.B) return;
  const int lane = threadIdx.x & 63;
  const int T = p.T, F = p.F;
  const long long n = (long long)T * F;
  float* spec = p.spec + b * n;
  const float* tg = p.tgradw + b * n;
  const float* fg = p.fgradw + b * n;
  float* phase = p.phase + b * n;
  extern __shared__ __attribute__((aligned(16))) u64 heap_top[];
  const size_t per_wave = (size_t)(p.heap_lds_cap + 1) + (size_t)(p.seg_cap + 1) / 2;     // u64 units: heap top, segment maxima
  u64* my_lds = heap_top + (size_t)wave * per_wave;
  const Heap H = {my_lds, reinterpret_cast<u64*>(p.heap + b * (n + 2)), p.heap_lds_cap};
  SegMax G;
  G.m = reinterpret_cast<float*>(my_lds + p.heap_lds_cap + 1);
  G.SL = 512 * ((n + 512LL * p.seg_cap - 1) / (512LL * p.seg_cap));
  if (G.SL < 512) G.SL = 512;
  G.S = (int)((n + G.SL - 1) / G.SL);
  const u64 anc_mask = chain_mask(lane);
  int* order = p.order ? p.order + b * n : nullptr;
  const float abstol = p.abstol;
  const float inv_F = 1.0f / (float)F;

  for (long long i = lane; i < n; i += 64) phase[i] = 0.0f;  // dgt.py:170

  float max_val;
  long long max_pos;
  seg_rebuild(spec, n, G, abstol, 0.f, false, lane, max_val, max_pos);  // :173-174 (+ the segment bounds)
  const float thr = max_val * p.tol;                                   // :177-178
  long long npops = 0;
  long long c_pop1 = 0, c_bubble = 0, c_sift = 0, c_nb = 0, c_push = 0, n_push = 0, s_depth = 0, hn_max = 0;
#define TICK() ((long long)__builtin_amdgcn_s_memtime())
  if (lane == 0) {
    H.store(0, pack_item(-max_val, (int)max_pos));  // :175
    spec[max_pos] = abstol;                         // :176
  }
  int hn = 1;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  while (max_val > abstol) {  // :179
    while (hn > 0) {          // :180
      hn = uni(hn);
      const long long t0 = PROF ? TICK() : 0;
      if (PROF) { s_depth += 31 - __clz((unsigned)hn | 1u); if (hn > hn_max) hn_max = hn; }
      // heappop, part 1: take the last entry off, read the root (heapq.py:51-56)
      const u64 top63 = H.top[lane >= 1 ? lane - 1 : 0];   // root + the first bubble round's subtree, one read
      const u64 last = H.load(hn - 1);     // usually deep in the global part: not needed before the leaf is known
      hn -= 1;
      int c;
      if (hn == 0) c = uni(item_idx(last));
      else c = __builtin_amdgcn_readlane(item_idx(top63), 1);   // the root always lives in LDS: no wait on `last`
      if (order && !PROF && lane == 0) order[npops] = c;
      ++npops;
      // frame / bin of c without an integer division (~25 dependent instructions on the pop's critical path):
      // float quotient, exact after one correction either way
      int col = (int)((float)c * inv_F);
      int row = c - col * F;
      if (row < 0) { row += F; col -= 1; }
      else if (row >= F) { row -= F; col += 1; }
      // request the neighbourhood now (lanes 0..3: next frame, previous frame, next bin, previous bin,
      // dgt.py:188-215); it does not depend on the heap repair below and arrives while that runs
      const int d = (lane == 0) ? F : (lane == 1) ? -F : (lane == 2) ? 1 : -1;
      const bool inb = (lane == 0) ? (col < T - 1) : (lane == 1) ? (col > 0) : (lane == 2) ? (row < F - 1)
                                                                                            : (lane == 3) && (row > 0);
      // Every lane loads (out-of-range neighbours and lanes >= 4 re-read bin c itself: the same cache lines), and
      // every loaded value is consumed outside any branch below.  A load the compiler has to treat as "maybe
      // still pending" at the loop's back edge makes it drain vmcnt at the top of the next pop -- which then
      // starts by sitting out the load of `last`, a deep heap entry, before it has issued anything else.
      const int nb = inb ? c + d : c;
      const float* gr = (lane < 2) ? fg : tg;
      const float s = fload(spec + nb);
      const float g_c = gr[c];
      const float g_n = gr[nb];
      const float pc = fload(phase + c);
      const long long t1 = PROF ? TICK() : 0;
      // heappop, part 2: bubble the smaller children up, drop `last` into the leaf, let it rise
      long long t2 = t1;
      if (hn > 0) {
        u64 leaf_old = 0;
        const int leaf = coop_bubble(H, hn, lane, anc_mask, leaf_old, top63);
        t2 = PROF ? TICK() : 0;
        // `last` goes into the leaf and rises while it is smaller than its parent (heapq.py:39-42).  The
        // parent of the leaf now holds the entry that just left the leaf, which is still in registers: in
        // the common case (`last` does not rise at all) no ancestor has to be read back.
        if (leaf == 0 || !(item_key(last) < item_key(leaf_old))) {
          if (lane == 0) H.store(leaf, last);
        } else {
          coop_siftdown(H, leaf, last, lane);
        }
      }
      const long long t3 = PROF ? TICK() : 0;
      const float half = (g_c + g_n) / 2.0f;
      const float new_phase = (lane & 1) ? pc - half : pc + half;
      const bool lv = inb && live(s, abstol, thr);     // inb is false on lanes >= 4
      if (lv) {
        phase[nb] = new_phase;
        spec[nb] = abstol;
      }
      const u64 lvmask = __ballot(lv);
      const long long t4 = PROF ? TICK() : 0;
      const u64 mine = pack_item(-s, nb);
      // heappush x (0..4), in lane order (heapq.py:45-48).
      // PUSH_BATCH (experiment, off by default): an entry that stays at its leaf changes nothing its successors look
      // at (their parents lie above the old end of the heap once it holds four entries), so all candidates test their
      // own parent in ONE round on their own lanes, the leading run of entries that stay put is stored at once, and
      // only from the first entry that does rise onwards the pushes take the cooperative sift one after the other.
      // Exact (PGHI suite green), but on dense spectra most pushed entries DO rise: the parent test is then paid on
      // top of the sift -- pushes 816 -> 1115 cycles per pop, 1024 dense clips 0.695 -> 0.743 s
      // (profiles/r03_pghi_kernels.md).
      int first_slow = 0;
      if (PUSH_BATCH && lvmask != 0 && hn >= 4) {
        const int rank = __builtin_popcountll(lvmask & ((1ull << lane) - 1ull));   // lanes 0..3 matter
        const int my_pos = hn + rank;
        const u64 par = H.load_if((my_pos - 1) >> 1, lv, 0);
        const bool rise = lv && (item_key(mine) < item_key(par));
        const u64 rmask = __ballot(rise);
        first_slow = rmask ? __builtin_ctzll(rmask) : 4;
        const bool fast = lv && lane < first_slow;
        if (fast) H.store(my_pos, mine);
        const int n_fast = __builtin_popcountll(lvmask & ((1ull << first_slow) - 1ull));
        hn += n_fast;
        n_push += n_fast;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= first_slow && ((lvmask >> q) & 1ull)) {
          const u64 item = readlane64(mine, q);
          coop_siftdown(H, hn, item, lane);
          ++hn;
          ++n_push;
        }
      }
      if (PROF) {
        const long long t5 = TICK();
        c_pop1 += t1 - t0; c_bubble += t2 - t1; c_sift += t3 - t2; c_nb += t4 - t3; c_push += t5 - t4;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");   // the scan below must not hit stale L1 lines
    // :216-219 reseed from the global max of what is left (lane-parallel scan)
    seg_reseed(spec, n, G, abstol, thr, lane, max_val, max_pos);
    if (lane == 0) {
      H.store(0, pack_item(-max_val, (int)max_pos));
      spec[max_pos] = abstol;
    }
    hn = 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  if (p.npops && lane == 0) p.npops[b] = npops;
  if (PROF && order && lane == 0 && b == 0) {
    long long* o = reinterpret_cast<long long*>(order);
    o[0] = npops; o[1] = c_pop1; o[2] = c_bubble; o[3] = c_sift; o[4] = c_nb; o[5] = c_push; o[6] = n_push; o[7] = s_depth; o[8] = hn_max;
  }
#undef TICK
}

// ---------------------------------------------------------------------------
// K14 offline, winner-bit variant of the wave-cooperative heap (opt-in: ACIDS_PGHI_KERNEL=wbit; an experiment kept
// runnable, NOT the default -- measured slower, see the end of this comment).
//
// Same array-embedded binary heap, same sift rules, same pop order -- what changes is how the pop finds its
// bubble-up path.  Every internal node of the top 17 levels carries one "winner" bit -- 1 iff heapq.py:33 would
// take the right child: it exists and not (left.key < right.key) -- so the path root -> leaf of a pop is read off
// the bits by ~50 scalar instructions (three LDS words: levels 0-5, 6-11, 12-16) BEFORE any heap entry is loaded.
// All entries the pop needs -- per level: the child that moves up, the grandchild that becomes its new value, the
// sibling it is compared with for the new bit -- are then requested in ONE parallel round (lane k = level k),
// where the cooperative kernel above resolves five levels per dependent round.  A push is one round too
// (ancestors and their siblings by index).  The bits are derived data, kept exact by these rules:
//   * a pop rewrites the bits of the nodes on its path above the slot `last` ends up in;
//   * a push rewrites the bits of the ancestors whose chain child changed (those it passed, plus one);
//   * a bit left pointing at a right child that was since taken off the end of the heap is recognised when the
//     path is read (the position equals the heap's size) and sends the path to the left sibling, a leaf;
// every bit is thus written when its node first gets a child and whenever a child's key changes (model-checked
// against CPython's heapq with heavily tied keys before it was written in HIP).  Levels below 17 (heaps beyond
// 262 143 entries) continue with plain child compares, one dependent round per level.
// Outcome (profiles/r02b_pghi_kernels.md, 1024 dense clips): 2 global round trips per pop instead of ~2.3, but 477
// instructions per pop against 360 -- and with one wave per SIMD a pop costs ~4.7 cycles per instruction whatever
// the memory does: 0.765 s against 0.706 s.  Requesting the first push's ancestors ahead of the pop's round made it
// slower still (the register allocator reuses the prefetch registers, which drains vmcnt early).
// ---------------------------------------------------------------------------
constexpr int WB_LEVELS = 17;                    // winner bits for internal nodes at levels 0..16
constexpr int WB_T1 = 2, WB_T2 = 2 + 128;        // u32 word offsets of the tiers: [t0: 2][t1: 64 x 2][t2: 4096]
constexpr int WB_WORDS = 2 + 128 + 4096;

__device__ __forceinline__ u64 rfl64(u64 v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// winner bit of internal node `node` (level <= 16) <- bit.  Called by single lanes; lanes of one wave may hit the
// same word (LDS atomics).
__device__ __forceinline__ void wb_write(unsigned* wb, int node, bool bit) {
  const unsigned q = (unsigned)node + 1u;
  const int lv = 31 - __clz(q);
  const unsigned o = q - (1u << lv);
  const int base_lv = lv < 6 ? 0 : (lv < 12 ? 6 : 12);
  const int d = lv - base_lv;
  const unsigned j = o >> d;                                  // tier word (the ancestor at the tier's first level)
  const unsigned local = (1u << d) + (o & ((1u << d) - 1u));  // heap-order index inside the tier subtree, 1-based
  const unsigned widx = lv < 6 ? (local >> 5) : (lv < 12 ? WB_T1 + 2 * j + (local >> 5) : WB_T2 + j);
  const unsigned m = 1u << (local & 31);
  __hip_atomic_fetch_and(wb + widx, ~m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (bit) __hip_atomic_fetch_or(wb + widx, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// follow the bits from the root: returns the 17 choices, first level in the top bit (scalar unit)
__device__ __forceinline__ unsigned wb_follow(const unsigned* wb, int n) {
  u64 w = rfl64(*reinterpret_cast<const u64*>(wb));
  unsigned i = 1;
#pragma unroll
  for (int k = 0; k < 6; ++k) i = 2 * i + (unsigned)((w >> i) & 1ull);
  const unsigned j1 = i - 64;
  unsigned i2 = 1, i3 = 1, j2 = j1 << 6;
  if (n > 63) {
    w = rfl64(*reinterpret_cast<const u64*>(wb + WB_T1 + 2 * j1));
#pragma unroll
    for (int k = 0; k < 6; ++k) i2 = 2 * i2 + (unsigned)((w >> i2) & 1ull);
    j2 = (j1 << 6) | (i2 - 64);
    if (n > 4095) {
      const unsigned w2 = (unsigned)__builtin_amdgcn_readfirstlane((int)wb[WB_T2 + j2]);
#pragma unroll
      for (int k = 0; k < 5; ++k) i3 = 2 * i3 + ((w2 >> i3) & 1u);
      return (j2 << 5) | (i3 - 32);
    }
    return j2 << 5;
  }
  return j1 << 11;
}

// three (two) entries per lane in one round: LDS part unconditionally, global part under one uniform test.  The
// scheduling barriers keep the loads back to back: left alone, the scheduler slips the first load's select between
// them and the wait that select needs turns one round trip into three.
__device__ __forceinline__ void wb_load3(const Heap& H, int pa, bool va, int pb, bool vb, int pc, bool vc, u64 dflt,
                                         u64& a, u64& b, u64& c) {
  const bool ga = va && pa >= H.cap, gb = vb && pb >= H.cap, gc = vc && pc >= H.cap;
  a = H.top[(va && !ga) ? pa : 0];
  b = H.top[(vb && !gb) ? pb : 0];
  c = H.top[(vc && !gc) ? pc : 0];
  if (__ballot(ga || gb || gc)) {
    const u64* qa = H.rest + (ga ? pa : 0);
    const u64* qb = H.rest + (gb ? pb : 0);
    const u64* qc = H.rest + (gc ? pc : 0);
    __builtin_amdgcn_sched_barrier(0);
    const u64 xa = gload(qa);
    const u64 xb = gload(qb);
    const u64 xc = gload(qc);
    __builtin_amdgcn_sched_barrier(0);
    a = ga ? xa : a;
    b = gb ? xb : b;
    c = gc ? xc : c;
  }
  a = va ? a : dflt;
  b = vb ? b : dflt;
  c = vc ? c : dflt;
}
__device__ __forceinline__ void wb_load2(const Heap& H, int pa, bool va, int pb, bool vb, u64& a, u64& b) {
  const bool ga = va && pa >= H.cap, gb = vb && pb >= H.cap;
  a = H.top[(va && !ga) ? pa : 0];
  b = H.top[(vb && !gb) ? pb : 0];
  if (__ballot(ga || gb)) {
    const u64* qa = H.rest + (ga ? pa : 0);
    const u64* qb = H.rest + (gb ? pb : 0);
    __builtin_amdgcn_sched_barrier(0);
    const u64 xa = gload(qa);
    const u64 xb = gload(qb);
    __builtin_amdgcn_sched_barrier(0);
    a = ga ? xa : a;
    b = gb ? xb : b;
  }
}

// heappush (heapq.py:45-48, 9-21): `item` goes to position `pos` (= the heap's size) and rises; bits of the
// ancestors whose chain child changed are rewritten
__device__ __forceinline__ void wb_push(const Heap& H, unsigned* wb, int pos, u64 item, int lane) {
  const unsigned q = (unsigned)pos + 1u;
  const int depth = 31 - __clz(q);                // number of ancestors (<= 25)
  const int sh = lane < 31 ? lane : 30;
  const int my_dst = (int)(q >> sh) - 1;          // lane j: the chain node below ancestor j+1 (j = 0: pos itself)
  const int my_anc = (int)(q >> (sh + 1)) - 1;    // lane j: ancestor j+1
  const bool act = lane < depth;
  const int sibp = (my_dst & 1) ? my_dst + 1 : my_dst - 1;
  const bool sib_ok = act && sibp <= pos;         // only pos's own right sibling can be missing
  u64 anc, sb;
  wb_load2(H, my_anc, act, sibp, sib_ok, anc, sb);
  const bool rises = act && (item_key(item) < item_key(anc));
  const u64 mask = __ballot(rises);
  const int m = (mask == ~0ull) ? 64 : __builtin_ctzll(~mask);  // item passes ancestors 1 .. m
  if (lane <= m && lane <= depth) H.store(my_dst, lane == m ? item : anc);
  // ancestor j+1 (lane j <= min(m, depth-1)): its chain child my_dst now holds `anc` (j < m) or the item (j == m)
  const int anc_level = depth - 1 - lane;
  if (act && lane <= m && anc_level < WB_LEVELS) {
    const float nk = item_key(lane < m ? anc : item);
    const bool bit = (my_dst & 1) ? (sib_ok && !(nk < item_key(sb))) : !(item_key(sb) < nk);
    wb_write(wb, my_anc, bit);
  }
}

__global__ __launch_bounds__(512) void pghi_hgi_offline_wbit_kernel(HgiParams p) {
  const int wave = threadIdx.x >> 6;
  const long long b = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (b >= p.B) return;
  const int lane = threadIdx.x & 63;
  const int T = p.T, F = p.F;
  const long long n = (long long)T * F;
  float* spec = p.spec + b * n;
  const float* tg = p.tgradw + b * n;
  const float* fg = p.fgradw + b * n;
  float* phase = p.phase + b * n;
  extern __shared__ __attribute__((aligned(16))) u64 heap_top[];
  const size_t per_wave = (size_t)(p.heap_lds_cap + 1) + (WB_WORDS + 1) / 2;      // u64 units
  u64* my_lds = heap_top + (size_t)wave * per_wave;
  const Heap H = {my_lds, reinterpret_cast<u64*>(p.heap + b * (n + 2)), p.heap_lds_cap};
  unsigned* wb = reinterpret_cast<unsigned*>(my_lds + p.heap_lds_cap + 1);
  int* order = p.order ? p.order + b * n : nullptr;
  const float abstol = p.abstol;
  const float inv_F = 1.0f / (float)F;
  const u64 kInf = (u64)0x7f800000u << 32;

  for (long long i = lane; i < n; i += 64) phase[i] = 0.0f;  // dgt.py:170

  float max_val;
  long long max_pos;
  clip_argmax(spec, n, abstol, 0.f, false, lane, max_val, max_pos);  // :173-174
  const float thr = max_val * p.tol;                                   // :177-178
  long long npops = 0;
  if (lane == 0) {
    H.store(0, pack_item(-max_val, (int)max_pos));  // :175
    spec[max_pos] = abstol;                         // :176
  }
  int hn = 1;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  while (max_val > abstol) {  // :179
    while (hn > 0) {          // :180
      hn = uni(hn);
      // heappop, part 1 (heapq.py:51-56): take the last entry off; the root is what is returned
      const int r = hn - 1;
      const u64 rootv = H.top[0];
      const u64 last = H.load(r);
      hn -= 1;
      const int c = (hn == 0) ? uni(item_idx(last)) : uni(item_idx(rootv));
      if (order && lane == 0) order[npops] = c;
      ++npops;
      int col = (int)((float)c * inv_F);
      int row = c - col * F;
      if (row < 0) { row += F; col -= 1; }
      else if (row >= F) { row -= F; col += 1; }
      // neighbourhood (lanes 0..3: next frame, previous frame, next bin, previous bin; dgt.py:188-215), requested now
      const int d = (lane == 0) ? F : (lane == 1) ? -F : (lane == 2) ? 1 : -1;
      const bool inb = (lane == 0) ? (col < T - 1) : (lane == 1) ? (col > 0) : (lane == 2) ? (row < F - 1)
                                                                                            : (lane == 3) && (row > 0);
      const int nb = inb ? c + d : c;
      const float* gr = (lane < 2) ? fg : tg;
      const float s = fload(spec + nb);
      const float g_c = gr[c];
      const float g_n = gr[nb];
      const float pc = fload(phase + c);

      if (hn > 0) {
        // heappop, part 2 (heapq.py:24-42): the path of smaller children from the root, read off the bits ...
        unsigned path = wb_follow(wb, hn);
        int plen = WB_LEVELS;                                   // levels described by `path`
        {
          int pos = (1 << WB_LEVELS) - 1 + (int)path;           // the level-17 node of the path
          // ... and, below level 17, by comparing the children (heaps beyond 2^18 - 1 entries only)
          while (pos < hn && 2 * pos + 1 < hn && plen < 31) {
            const int cl = 2 * pos + 1;
            const float kl = item_key(H.load(cl));
            const bool has_r = cl + 1 < hn;
            const float kr = has_r ? item_key(H.load(cl + 1)) : 0.f;
            const unsigned bsel = (has_r && !(kl < kr)) ? 1u : 0u;
            path = (path << 1) | (unsigned)uni((int)bsel);
            pos = cl + (int)(path & 1u);
            ++plen;
          }
        }
        // lane k = level k: p_k = (2^k - 1) + (first k choices)
        // A bit may still point at a right child that has since been taken off the end of the heap (position hn,
        // even): its left sibling hn - 1 is then the only child, and a leaf -- the path ends there.  (Nothing else
        // can be stale: the bit is rewritten as soon as position hn is filled again.)
        const int k1 = lane + 1, k2 = lane + 2;
        const int gone = (hn & 1) ? -1 : hn;
        int p0 = lane <= plen ? (1 << lane) - 1 + (int)(path >> (plen - lane)) : 0x7fffffff;
        int p1 = k1 <= plen ? (1 << k1) - 1 + (int)(path >> (plen - k1)) : 0x7fffffff;
        int p2 = k2 <= plen ? (1 << k2) - 1 + (int)(path >> (plen - k2)) : 0x7fffffff;
        p0 = p0 == gone ? hn - 1 : p0;
        p1 = p1 == gone ? hn - 1 : p1;
        p2 = p2 == gone ? hn - 1 : p2;
        const int L = __builtin_popcountll(__ballot(p0 < hn)) - 1;          // the path ends at level L (a prefix is valid)
        const bool v1 = p1 < hn, v2 = p2 < hn;
        const int sb1 = (p1 & 1) ? p1 + 1 : p1 - 1;                          // sibling of the child that moves up
        const bool vs = v1 && sb1 < hn;
        u64 X1, X2, S1;
        wb_load3(H, p1, v1, p2, v2, sb1, vs, kInf, X1, X2, S1);
        // `last` goes into the leaf p_L and rises while it is smaller than its parent (heapq.py:39-42): past the
        // entry that moved into p_{L-1} (the old p_L), p_{L-2}, ...  Levels below where it stops keep their entries.
        const u64 R = __ballot(v1 && item_key(last) < item_key(X1));          // bit k: passes the old entry of p_{k+1}
        const u64 Z = ~R & ((1ull << L) - 1ull);
        const int m = Z ? (L - 1) - (63 - __builtin_clzll(Z)) : L;
        const int Lp = L - m;                                                  // `last` ends up in p_{L'}
        if (lane < Lp) H.store(p0, X1);
        else if (lane == Lp) H.store(p0, last);
        if (lane < Lp && lane < WB_LEVELS) {
          const float nk = item_key(lane + 1 < Lp ? X2 : last);               // new entry of the chain child p_{k+1}
          const bool bit = (p1 & 1) ? (vs && !(nk < item_key(S1))) : !(item_key(S1) < nk);
          wb_write(wb, p0, bit);
        }
      }
      const float half = (g_c + g_n) / 2.0f;
      const float new_phase = (lane & 1) ? pc - half : pc + half;
      const bool lv = inb && live(s, abstol, thr);     // inb is false on lanes >= 4
      if (lv) {
        phase[nb] = new_phase;
        spec[nb] = abstol;
      }
      const u64 lvmask = __ballot(lv);
      const u64 mine = pack_item(-s, nb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((lvmask >> q) & 1ull) {
          const u64 item = readlane64(mine, q);
          wb_push(H, wb, hn, item, lane);      // heappush (heapq.py:45-48)
          ++hn;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");   // the scan below must not hit stale L1 lines
    // :216-219 reseed from the global max of what is left (lane-parallel scan)
    clip_argmax(spec, n, abstol, thr, true, lane, max_val, max_pos);
    if (lane == 0) {
      H.store(0, pack_item(-max_val, (int)max_pos));
      spec[max_pos] = abstol;
    }
    hn = 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  if (p.npops && lane == 0) p.npops[b] = npops;
}

// ---------------------------------------------------------------------------
// realtime: gradients on the (R = n+2, F) stack [2 history frames ; n new frames]
// (dgt.py:378-397).  The reference's uninitialised time-border rows are defined as 0.
// ---------------------------------------------------------------------------
struct RtParams {
  const float* mag_hist;    // (S, 2, F)
  const float* mag;         // (S, n, F)
  const float* prev_phase;  // (S, F)
  const float* noise;       // (S, n, F): caller's draws, or the workspace array the gradient kernel fills (seeded mode)
  float* noise_gen;         // seeded mode: where the gradient kernel writes its draws (== noise), else null
  unsigned* rng_state;      // seeded mode: {seed lo, seed hi, step counter, unused}; the heap kernel bumps the counter
  float* phase_out;         // (S, n, F)
  float* spec;              // (S, R, F) work
  float* hist;              // (S, R, F) work (untouched copy)
  float* tgradw;            // (S, R, F) work
  float* fgradw;            // (S, R, F) work
  float* phase;             // (S, R, F) work
  HeapItem* heap;           // (S, 4F + 8)
  int S, n, F, n_fft, hop;
  float gamma, tol, eps;
  int lds_floats_per_wave;  // cooperative kernel: LDS floats per stream (several streams per workgroup)
  // rank fast path (pghi_rt_rank_kernel -> pghi_hgi_rt_coop_kernel): per (stream, new frame) the 2F candidates of the
  // frame's flood -- rows f-1 and f -- sorted by magnitude: ent_of_rank[2F], rank_of_ent[2F] (u16 each), then one bit per
  // rank: "same magnitude as the rank before".  Null: heap path only.
  unsigned short* ranks;
  long long rank_stride;    // u16 elements per (stream, frame) record
  int scan_path;            // 1: try the wavefront-parallel closure (rt_scan_frame) before the rank / heap floods
};

// one record: u16 ent_of_rank[2F], u16 rank_of_ent[2F], u32 same_as_previous[ceil(2F / 32)] (bit r: the magnitude at rank r
// equals the one at rank r - 1), padded to 16 bytes
__host__ __device__ inline long long rt_rank_stride_u16(int F) {
  return ((8LL * F + 4LL * ((2 * F + 31) / 32) + 15) & ~15LL) / 2;
}

__device__ __forceinline__ float rt_mag(const RtParams& p, int s, int j, int k) {
  const float v = (j < 2) ? p.mag_hist[((long long)s * 2 + j) * p.F + k] : p.mag[((long long)s * p.n + (j - 2)) * p.F + k];
  return fmaxf(v, p.eps);
}

// Standard-normal draws for the bins at or below the tolerance (dgt.py:404-405: torch.randn_like), generated on the
// device so that a captured streaming step needs no generator launch: Philox-4x32-10 keyed by the session's seed, counter
// = (element index, step counter), Box-Muller on two of the four outputs.  Statistical parity only (the reference's
// draws are torch's; tests/test_stream_quant_gpu.py checks mean / variance / tails / step-to-step independence).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned long long index, unsigned step, const unsigned* state) {
  unsigned r[4];
  philox4x32_10((unsigned)index, (unsigned)(index >> 32), step, 0x5eedu, state[0], state[1], r);
  const float u1 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);    // (0, 1]
  const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);             // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * __builtin_amdgcn_cosf(u2);             // v_cos_f32 takes revolutions
}

__global__ __launch_bounds__(256) void pghi_grad_rt_kernel(RtParams p) {
  const float fmul = p.gamma / (float)((long long)p.hop * (long long)p.n_fft);
  const float fstep = ((float)(2.0 * 3.14159265358979323846) * (float)p.hop) / (float)p.n_fft;
  const float pi_f = (float)3.14159265358979323846;
  const int R = p.n + 2;
  const long long per = (long long)R * p.F;
  const long long total = (long long)p.S * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(i / per);
    const long long r = i - (long long)s * per;
    const int j = (int)(r / p.F), k = (int)(r - (long long)j * p.F);
    const int kr = k + 1 < p.F ? k + 1 : p.F - 1, kl = k > 0 ? k - 1 : 0;
    const float c = rt_mag(p, s, j, k);
    const float right = logf(rt_mag(p, s, j, kr));
    const float left = logf(rt_mag(p, s, j, kl));
    const float nxt = (j + 1 < R) ? logf(rt_mag(p, s, j + 1, k)) : 0.0f;
    const float prv = (j > 0) ? logf(rt_mag(p, s, j - 1, k)) : 0.0f;
    const float dxdw = (right - left) / 2.0f;                          // :393
    const float dxdt = (3.0f * nxt - 4.0f * logf(c) + prv) / 2.0f;     // :394
    p.fgradw[i] = dxdw / fmul + fstep * (float)k;                      // :395
    p.tgradw[i] = (-fmul) * dxdt + pi_f;                               // :396
    p.spec[i] = c;
    p.hist[i] = c;
    if (p.noise_gen && j >= 2)      // one draw per new bin (rows 2 .. n + 1), same indexing as a caller-supplied array
      p.noise_gen[(long long)s * p.n * p.F + (long long)(j - 2) * p.F + k] =
          philox_normal((unsigned long long)s * p.n * p.F + (unsigned long long)(j - 2) * p.F + k, p.rng_state[2], p.rng_state);
  }
}

// dgt.py:399-466, one wave per stream
__global__ __launch_bounds__(64) void pghi_hgi_rt_kernel(RtParams p) {
  if (p.rng_state && blockIdx.x == 0 && threadIdx.x == 0) p.rng_state[2] += 1u;   // next step draws from a new counter (the gradient kernel has finished)
  const int s = blockIdx.x;
  if (s >= p.S) return;
  const int lane = threadIdx.x;
  const int F = p.F, R = p.n + 2;
  const long long n = (long long)R * F;
  float* spec = p.spec + (long long)s * n;
  const float* hist = p.hist + (long long)s * n;
  const float* tgw = p.tgradw + (long long)s * n;
  const float* fgw = p.fgradw + (long long)s * n;
  float* phase = p.phase + (long long)s * n;
  HeapItem* heap = p.heap + (long long)s * (4LL * F + 8);

  // :400 abstol = clamp(tol * max(spec), eps)
  float smax = -1.0f;
  for (long long i = lane; i < n; i += 64) smax = fmaxf(smax, spec[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 64));
  float abstol = p.tol * smax;
  if (abstol < p.eps) abstol = p.eps;

  // :402-405 initial phase: row 0 zeros, row 1 previous phase, rows >= 2 zero above abstol else noise
  for (long long i = lane; i < n; i += 64) {
    float v;
    if (i < F) v = 0.0f;
    else if (i < 2LL * F) v = p.prev_phase[(long long)s * F + (i - F)];
    else v = (spec[i] > abstol) ? 0.0f : p.noise[(long long)s * p.n * F + (i - 2LL * F)];
    phase[i] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __threadfence_block();

  // the reference front-pads the gradients with two zero rows (:408-410): "row r" there is row r-2 here
#define TG(r, k) ((r) >= 2 ? tgw[(long long)((r)-2) * F + (k)] : 0.0f)
#define FG(r, k) ((r) >= 2 ? fgw[(long long)((r)-2) * F + (k)] : 0.0f)
  for (int f = 2; f < R; ++f) {  // :413
    float* row = spec + (long long)f * F;
    float max_val = -1.0f;
    long long max_k = F;
    for (int k = lane; k < F; k += 64) {
      const float v = row[k];
      if (v > max_val) {
        max_val = v;
        max_k = k;
      }
    }
    wave_argmax(max_val, max_k);
    if (max_val <= abstol) continue;  // :416-417
    if (lane == 0) {
      int hn = 0;
      heap[0].key = -max_val;  // :427 the seed is NOT marked visited
      heap[0].idx = f * F + (int)max_k;
      hn = 1;
      for (int k = 0; k < F; ++k) {  // :428-430
        const float hv = hist[(long long)(f - 1) * F + k];
        if (hv > abstol) h_push(heap, hn, -hv, (f - 1) * F + k);
      }
      while (max_val > abstol) {  // :433
        while (hn > 0) {
          const HeapItem it = h_pop(heap, hn);
          const int r = it.idx / F, k = it.idx - r * F;
          if (r == f - 1) {  // :436-443 propagate in time
            const float sv = row[k];
            if (sv > abstol) {
              phase[(long long)f * F + k] = phase[(long long)(f - 1) * F + k] + 0.5f * (TG(f - 1, k) + TG(f, k));
              h_push(heap, hn, -sv, f * F + k);
              row[k] = abstol;
            }
          }
          if (r == f) {  // :444-460 propagate in frequency
            if (k + 1 < F) {
              const float sv = row[k + 1];
              if (sv > abstol) {
                phase[(long long)f * F + k + 1] = phase[(long long)f * F + k] + 0.5f * (FG(f, k) + FG(f, k + 1));
                h_push(heap, hn, -sv, f * F + k + 1);
                row[k + 1] = abstol;
              }
            }
            if (k - 1 > 0) {  // bin 0 is never reached downward (:453)
              const float sv = row[k - 1];
              if (sv > abstol) {
                phase[(long long)f * F + k - 1] = phase[(long long)f * F + k] - 0.5f * (FG(f, k) + FG(f, k - 1));
                h_push(heap, hn, -sv, f * F + k - 1);
                row[k - 1] = abstol;
              }
            }
          }
        }
        // :461-465 reseed inside the frame
        max_val = row[0];
        int mk = 0;
        for (int k = 1; k < F; ++k)
          if (row[k] > max_val) {
            max_val = row[k];
            mk = k;
          }
        h_push(heap, hn, -max_val, f * F + mk);
        row[mk] = abstol;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __threadfence_block();
  }
#undef TG
#undef FG
  for (long long i = lane; i < (long long)p.n * F; i += 64) p.phase_out[(long long)s * p.n * F + i] = phase[2LL * F + i];
}

// Same algorithm with the per-frame state in LDS: a frame's heap holds at most ~2F entries and touches only
// two rows of each array, so the whole frame step runs out of the CU's LDS (one wave per stream, F <= 1025).
__global__ __launch_bounds__(64) void pghi_hgi_rt_lds_kernel(RtParams p) {
  if (p.rng_state && blockIdx.x == 0 && threadIdx.x == 0) p.rng_state[2] += 1u;   // next step draws from a new counter (the gradient kernel has finished)
  extern __shared__ __attribute__((aligned(16))) float rt_smem[];
  const int s = blockIdx.x;
  if (s >= p.S) return;
  const int lane = threadIdx.x;
  const int F = p.F, R = p.n + 2;
  const long long n = (long long)R * F;
  const float* spec = p.spec + (long long)s * n;
  const float* tgw = p.tgradw + (long long)s * n;
  const float* fgw = p.fgradw + (long long)s * n;
  float* phase = p.phase + (long long)s * n;
  float* srow = rt_smem;       // working copy of spectrogram row f
  float* hrow = srow + F;      // untouched row f-1 (spectrogram_history, dgt.py:411)
  float* ph0 = hrow + F;       // phase row f-1
  float* ph1 = ph0 + F;        // phase row f
  float* tg0 = ph1 + F;        // padded tgradw rows f-1 and f  (= rows f-3, f-2 of the unpadded array, or 0)
  float* tg1 = tg0 + F;
  float* fg1 = tg1 + F;        // padded fgradw row f
  HeapItem* heap = reinterpret_cast<HeapItem*>(fg1 + F + (F & 1));

  float smax = -1.0f;
  for (long long i = lane; i < n; i += 64) smax = fmaxf(smax, spec[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 64));
  float abstol = p.tol * smax;  // :400
  if (abstol < p.eps) abstol = p.eps;

  for (int k = lane; k < F; k += 64) {  // :402-403 rows 0 and 1 of the phase array
    phase[k] = 0.0f;
    phase[F + k] = p.prev_phase[(long long)s * F + k];
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");

  for (int f = 2; f < R; ++f) {  // :413
    float max_val = -1.0f;
    long long max_k = F;
    for (int k = lane; k < F; k += 64) {
      const float v = spec[(long long)f * F + k];
      srow[k] = v;
      hrow[k] = spec[(long long)(f - 1) * F + k];
      ph0[k] = phase[(long long)(f - 1) * F + k];
      ph1[k] = (v > abstol) ? 0.0f : p.noise[(long long)s * p.n * F + (long long)(f - 2) * F + k];  // :404-405
      tg0[k] = (f - 1 >= 2) ? tgw[(long long)(f - 3) * F + k] : 0.0f;  // :408-410 two-row front padding
      tg1[k] = tgw[(long long)(f - 2) * F + k];
      fg1[k] = fgw[(long long)(f - 2) * F + k];
      if (v > max_val) {
        max_val = v;
        max_k = k;
      }
    }
    wave_argmax(max_val, max_k);
    __syncthreads();
    if (max_val > abstol && lane == 0) {  // :416-417
      int hn = 0;
      heap[0].key = -max_val;  // :427 the seed is NOT marked visited
      heap[0].idx = F + (int)max_k;  // idx = (row == f ? F : 0) + bin
      hn = 1;
      for (int k = 0; k < F; ++k) {  // :428-430
        const float hv = hrow[k];
        if (hv > abstol) h_push(heap, hn, -hv, k);
      }
      while (max_val > abstol) {  // :433
        while (hn > 0) {
          const HeapItem it = h_pop(heap, hn);
          const bool cur_row = it.idx >= F;
          const int k = cur_row ? it.idx - F : it.idx;
          if (!cur_row) {  // :436-443 propagate in time
            const float sv = srow[k];
            if (sv > abstol) {
              ph1[k] = ph0[k] + 0.5f * (tg0[k] + tg1[k]);
              h_push(heap, hn, -sv, F + k);
              srow[k] = abstol;
            }
          } else {  // :444-460 propagate in frequency
            if (k + 1 < F) {
              const float sv = srow[k + 1];
              if (sv > abstol) {
                ph1[k + 1] = ph1[k] + 0.5f * (fg1[k] + fg1[k + 1]);
                h_push(heap, hn, -sv, F + k + 1);
                srow[k + 1] = abstol;
              }
            }
            if (k - 1 > 0) {  // bin 0 is never reached downward (:453)
              const float sv = srow[k - 1];
              if (sv > abstol) {
                ph1[k - 1] = ph1[k] - 0.5f * (fg1[k] + fg1[k - 1]);
                h_push(heap, hn, -sv, F + k - 1);
                srow[k - 1] = abstol;
              }
            }
          }
        }
        // :461-465 reseed inside the frame
        max_val = srow[0];
        int mk = 0;
        for (int k = 1; k < F; ++k)
          if (srow[k] > max_val) {
            max_val = srow[k];
            mk = k;
          }
        h_push(heap, hn, -max_val, F + mk);
        srow[mk] = abstol;
      }
    }
    __syncthreads();
    for (int k = lane; k < F; k += 64) {
      const float v = ph1[k];
      phase[(long long)f * F + k] = v;
      p.phase_out[(long long)s * p.n * F + (long long)(f - 2) * F + k] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __syncthreads();
  }
}

#ifdef AT_DEV_SWITCHES
__device__ unsigned g_rt_stats[8];     // dev builds: frames, scan successes, resolution rounds, declined: reseed / tie
#define RT_STAT(i, v) do { if (lane == 0) atomicAdd(&g_rt_stats[i], (unsigned)(v)); } while (0)
#else
#define RT_STAT(i, v) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// Realtime flood of ONE frame without a queue (round 5): the wavefront-parallel closure.
//
// The frame's flood (dgt.py:413-465) runs on a two-row strip: every live bin of row f-1 is a SOURCE in the heap from the
// start (key a_k = its magnitude, phase known), row f is unvisited (key b_k); a popped source (f-1, k) visits (f, k), a
// popped (f, k) visits (f, k+1) and -- unless that would be bin 0 (:453) -- (f, k-1); "visit" = take the phase from the
// popping entry, push, mark.  With DISTINCT keys the order of pops does not depend on the queue, and neither does the
// one thing the phases depend on: WHICH neighbour reaches a bin first.  An entry pops at the level
//     lev(e) = min(key(e), level at which e was pushed)        (the running minimum of the popped keys when e leaves)
// and entries leave in order of decreasing level, so bin k is visited at R(k) = max(a_k, lev(f, k-1), lev(f, k+1)) by
// whichever of the three attains the maximum, and lev(f, k) = min(b_k, R(k)).  On a path graph the closure is two
// directional scans of clamp functions x -> min(b_k, max(a_k, x)) (a composition of clamps is a clamp: a parallel prefix
// over the lanes), and the phases follow the parent pointers: chains along the row, each bin ONE float addition onto its
// parent's final phase -- the same additions as the heap flood, so the same bits.
//
// The unmarked seed (:427: the frame maximum is pushed first and NOT marked visited) is an extra entry S at bin kmax that
// pops at level b_kmax whatever happens to the bin itself, and hands its neighbours the phase the bin holds AT THAT TIME:
// the visited one if the bin's own source is larger than b_kmax, the initial one otherwise (the bin is then visited
// later, possibly by one of S's own children).
//
// Declined (returns false, nothing written; the caller runs the rank / heap flood): islands of several live bins that
// nothing reaches (the reference reseeds, :461-465 -- onsets in sparse spectra; a one-bin island just keeps its initial
// phase), two candidates for a bin's parent with EQUAL levels (tied magnitudes that actually compete: only the heap
// knows their order), the seed's bin visited at exactly the seed's level.  Ties that never meet are harmless.
//
// Lane l owns the CONTIGUOUS bins [l C, (l + 1) C), C = ceil(F / 64) <= CM (compile-time: every loop below is unrolled
// and predicated): the row is read from LDS once, the four scan walks, the parent choice and the rounds along the parent
// chains run on registers, chunk boundaries cross lanes by shuffles.  (A first version kept the arrivals and the chains
// in LDS: 15.6 us per frame against 10.5.)
template <int CM>
__device__ bool rt_scan_frame(const int F, const int lane, const float abstol, const int kmax, const float* srow,
                                   const float* hrow, const float* ph0, float* ph1, const float* tg0, const float* tg1,
                                   const float* fg1, float* scratch) {
  const float NEG = -__builtin_inff(), POS = __builtin_inff();
  int* par = reinterpret_cast<int*>(scratch);             // parent codes, for the island test across lane boundaries
  auto sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto apply = [](float lo, float hi, float v) { return fminf(hi, fmaxf(lo, v)); };
  const int C = (F + 63) >> 6;
  const int k0 = lane * C < F ? lane * C : F;
  const int k1 = k0 + C < F ? k0 + C : F;
  const int nb = k1 - k0;                                // bins this lane owns (0 .. C)
  float b_[CM], c_[CM], lo_[CM], hi_[CM], fg_[CM];
#pragma unroll
  for (int i = 0; i < CM; ++i) {
    const bool in = i < nb;
    const int k = in ? k0 + i : 0;
    const float b = srow[k], a = hrow[k];
    fg_[i] = fg1[k];
    b_[i] = b;
    c_[i] = a > abstol ? a : NEG;
    if (!in) {                                           // identity: the chunk ends before CM bins
      lo_[i] = NEG;
      hi_[i] = POS;
      b_[i] = NEG;
    } else if (k == kmax) {
      lo_[i] = hi_[i] = b;
    } else if (!(b > abstol)) {
      lo_[i] = hi_[i] = NEG;
    } else {
      hi_[i] = b;
      lo_[i] = fminf(c_[i], b);
    }
  }
  float x_[CM], y_[CM];
  {
    float lo = NEG, hi = POS;
#pragma unroll
    for (int i = 0; i < CM; ++i) {
      lo = apply(lo_[i], hi_[i], lo);
      hi = apply(lo_[i], hi_[i], hi);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float plo = __shfl_up(lo, d, 64), phi = __shfl_up(hi, d, 64);
      if (lane >= d) {
        const float nlo = apply(lo, hi, plo), nhi = apply(lo, hi, phi);
        lo = nlo;
        hi = nhi;
      }
    }
    float arr = __shfl_up(lo, 1, 64);
    if (lane == 0) arr = NEG;
#pragma unroll
    for (int i = 0; i < CM; ++i) {
      x_[i] = arr;
      arr = apply(lo_[i], hi_[i], arr);
    }
  }
  {
    float lo = NEG, hi = POS;
#pragma unroll
    for (int i = CM - 1; i >= 0; --i) {
      lo = apply(lo_[i], hi_[i], lo);
      hi = apply(lo_[i], hi_[i], hi);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float plo = __shfl_down(lo, d, 64), phi = __shfl_down(hi, d, 64);
      if (lane + d < 64) {
        const float nlo = apply(lo, hi, plo), nhi = apply(lo, hi, phi);
        lo = nlo;
        hi = nhi;
      }
    }
    float arr = __shfl_down(lo, 1, 64);
    if (lane == 63) arr = NEG;
#pragma unroll
    for (int i = CM - 1; i >= 0; --i) {
      y_[i] = (k0 + i >= 1) ? arr : NEG;                 // bin 0 is never reached downward
      arr = apply(lo_[i], hi_[i], arr);
    }
  }
  // ---- parents
  const float bmax = srow[kmax];
  const float amax = hrow[kmax];
  const float cmax = amax > abstol ? amax : NEG;
  const float b_below = kmax - 1 >= 1 ? srow[kmax - 1] : 0.0f;          // the seed's neighbours, as S's children
  const float b_above = (kmax >= 1 && kmax + 1 < F) ? srow[kmax + 1] : 0.0f;
  bool bad = false, bad_reseed = false, seed_tie = false;
  int code_[CM];
#pragma unroll
  for (int i = 0; i < CM; ++i) {
    const int k = k0 + i;
    int code = 0;
    if (i < nb && b_[i] > abstol) {
      const float c = c_[i];
      float x = x_[i], y = y_[i];
      if (k == kmax) {
        if (b_below > abstol) x = b_below;
        if (b_above > abstol) y = b_above;
      }
      const float m = fmaxf(c, fmaxf(x, y));
      if ((int)(c == m) + (int)(x == m) + (int)(y == m) > 1 && m != NEG) bad = true;
      // the seed's bin visited at the very level S pops at (its source, or a neighbour, ties with the frame maximum):
      // which phase S hands on is the heap's to say -- and matters only if S has a child (below)
      if (k == kmax && m == bmax) seed_tie = true;
      code = m == NEG ? 4 : (c == m ? 1 : (x == m ? 2 : 3));
    }
    code_[i] = code;
    if (i < nb) par[k] = code;
  }
  sync();
  // S's children: the bins next to the seed whose parent is the seed's side.  A held frame (row f equal to row f-1: a
  // stationary signal, hop-periodic tones) ties every source with its own bin -- harmless everywhere (a source only ever
  // visits its own bin) except at the seed, and there only if somebody takes S's phase.
  // ... and only if some source is LARGER than the frame maximum: S is pushed first (:427) and an entry only rises past an
  // equal key on a strict comparison (heapq.py:9-21), so with no larger source S is still the root when the flood starts and
  // is its very first pop -- it hands on the initial phase, whatever ties with it (a held frame: the row maxima are equal).
  if (__ballot(seed_tie) != 0 &&
      ((kmax + 1 < F && par[kmax + 1] == 2) || (kmax - 1 >= 1 && par[kmax - 1] == 3))) {
    float src_max = NEG;
#pragma unroll
    for (int i = 0; i < CM; ++i)
      if (i < nb) src_max = fmaxf(src_max, c_[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) src_max = fmaxf(src_max, __shfl_xor(src_max, o, 64));
    if (src_max > bmax) bad = true;
  }
#pragma unroll
  for (int i = 0; i < CM; ++i) {
    const int k = k0 + i;
    if (i < nb && code_[i] == 4 && ((k >= 1 && par[k - 1] == 4) || (k + 1 < F && par[k + 1] == 4))) bad = bad_reseed = true;
  }
  RT_STAT(0, 1);
  if (__ballot(bad) != 0) {
    RT_STAT(__ballot(bad_reseed) != 0 ? 3 : 4, 1);
    return false;
  }
  const float phi_s = cmax > bmax ? ph0[kmax] + 0.5f * (tg0[kmax] + tg1[kmax]) : ph1[kmax];
  // ---- phases: sources at once, then the chains; a bin adds its own step onto its parent's FINAL phase, nothing else
  float val_[CM];
  bool done_[CM];
#pragma unroll
  for (int i = 0; i < CM; ++i) {
    const int k = i < nb ? k0 + i : 0;
    done_[i] = false;
    val_[i] = 0.0f;
    if (code_[i] == 1) {
      val_[i] = ph0[k] + 0.5f * (tg0[k] + tg1[k]);
      done_[i] = true;
    }
  }
  const float fg_left = fg1[k0 >= 1 ? k0 - 1 : 0];       // the neighbours' frequency gradients across the chunk boundaries
  const float fg_right = fg1[k1 < F ? k1 : F - 1];
  for (int round = 0; round <= 64; ++round) {
    // what the lane to the left / right holds at the boundary: its last / first bin
    float last_val = 0.0f, first_val = val_[0];
    bool last_done = false, first_done = done_[0];
#pragma unroll
    for (int i = 0; i < CM; ++i)
      if (i == nb - 1) {
        last_val = val_[i];
        last_done = done_[i];
      }
    const float lv = __shfl_up(last_val, 1, 64), rv = __shfl_down(first_val, 1, 64);
    const bool ld = __shfl_up((int)last_done, 1, 64) != 0 && lane >= 1;
    const bool rd = __shfl_down((int)first_done, 1, 64) != 0 && lane < 63;
    bool pending = false;
#pragma unroll
    for (int i = 0; i < CM; ++i) {
      if (code_[i] == 2 && !done_[i]) {
        const int j = k0 + i - 1;
        const bool ready = j == kmax || (i == 0 ? ld : done_[i > 0 ? i - 1 : 0]);
        if (ready) {
          const float pp = j == kmax ? phi_s : (i == 0 ? lv : val_[i > 0 ? i - 1 : 0]);
          val_[i] = pp + 0.5f * ((i == 0 ? fg_left : fg_[i > 0 ? i - 1 : 0]) + fg_[i]);
          done_[i] = true;
        } else {
          pending = true;
        }
      }
    }
#pragma unroll
    for (int i = CM - 1; i >= 0; --i) {
      if (code_[i] == 3 && !done_[i]) {
        const int j = k0 + i + 1;
        const bool at_end = i == nb - 1;                 // the parent is the right-hand lane's first bin
        const bool ready = j == kmax || (at_end ? rd : done_[i + 1 < CM ? i + 1 : CM - 1]);
        if (ready) {
          const float pp = j == kmax ? phi_s : (at_end ? rv : val_[i + 1 < CM ? i + 1 : CM - 1]);
          val_[i] = pp - 0.5f * ((at_end ? fg_right : fg_[i + 1 < CM ? i + 1 : CM - 1]) + fg_[i]);
          done_[i] = true;
        } else {
          pending = true;
        }
      }
    }
    RT_STAT(2, 1);
    if (__ballot(pending) == 0) break;
  }
#pragma unroll
  for (int i = 0; i < CM; ++i)
    if (i < nb && code_[i] >= 1 && code_[i] <= 3) ph1[k0 + i] = val_[i];
  sync();
  RT_STAT(1, 1);
  return true;
}

// Rank pre-pass of the realtime flood.  A frame's flood (dgt.py:413-465) pops, in descending magnitude, entries that are
// known before it starts: the bins of row f-1 (pushed up front) and the bins of row f (pushed as the flood reaches
// them).  When no two of those magnitudes are equal, WHICH priority queue hands them out is immaterial -- the pop order
// is the strict order of the keys -- and a bitmap over the ranks replaces the heap: pop = find-first-set, push = set a
// bit.  Ranks are a sort, and a sort is parallel: one workgroup per (stream, frame) on the chip the one-wave-per-stream
// flood leaves idle.  Equal magnitudes are marked per rank (`same_as_previous`); what the flood does about them is in
// pghi_hgi_rt_coop_kernel.
__global__ __launch_bounds__(256) void pghi_rt_rank_kernel(RtParams p, int n2) {
  extern __shared__ __attribute__((aligned(16))) u64 rk_items[];
  const int F = p.F, R = p.n + 2, E = 2 * F;
  const int s = blockIdx.x / p.n, fr = blockIdx.x - s * p.n;          // new frame fr: rows fr + 1 (f - 1) and fr + 2 (f)
  const float* rows = p.spec + ((long long)s * R + fr + 1) * F;       // 2F consecutive floats: row f-1, then row f
  for (int i = threadIdx.x; i < n2; i += blockDim.x)
    rk_items[i] = i < E ? (((u64)(~__float_as_uint(rows[i]))) << 32) | (unsigned)i : ~0ull;   // ascending = descending magnitude
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const int x = i ^ j;
        if (x > i) {
          const u64 a = rk_items[i], b = rk_items[x];
          if ((a > b) == ((i & k) == 0)) {
            rk_items[i] = b;
            rk_items[x] = a;
          }
        }
      }
      __syncthreads();
    }
  unsigned short* rec = p.ranks + ((long long)s * p.n + fr) * p.rank_stride;
  unsigned* sp = reinterpret_cast<unsigned*>(rec + 2 * E);
  for (int r = threadIdx.x; r < E; r += blockDim.x) {
    const u64 it = rk_items[r];
    const unsigned e = (unsigned)it & 0xffffu;
    rec[r] = (unsigned short)e;
    rec[E + e] = (unsigned short)r;
  }
  // same-as-previous bits, one thread per word (magnitudes are compared as the bit patterns they were sorted by)
  for (int w = threadIdx.x; w < (E + 31) / 32; w += blockDim.x) {
    unsigned bits = 0u;
    for (int b = 0; b < 32; ++b) {
      const int r = 32 * w + b;
      if (r >= 1 && r < E && (unsigned)(rk_items[r] >> 32) == (unsigned)(rk_items[r - 1] >> 32)) bits |= 1u << b;
    }
    sp[w] = bits;
  }
}

// The same frame step with the wave-cooperative heap of the offline kernel (coop_bubble / coop_siftdown), the
// whole heap in LDS.  A frame pops up to 2F entries from a heap of ~1000: done by one lane that is ~10 dependent
// LDS round trips down and a few up per pop (~3600 cycles); the cooperative pop resolves five levels per round
// with one wide LDS gather.  Same binary heap, same sift rules, same pop order.
__global__ __launch_bounds__(256) void pghi_hgi_rt_coop_kernel(RtParams p) {
  if (p.rng_state && blockIdx.x == 0 && threadIdx.x == 0) p.rng_state[2] += 1u;   // next step draws from a new counter (the gradient kernel has finished)
  extern __shared__ __attribute__((aligned(16))) float rt_smem_all[];
  // one wave per stream, 1 to 4 independent waves per workgroup (see pghi_hgi_offline_coop_kernel)
  const int wave = threadIdx.x >> 6;
  const int s = blockIdx.x * (blockDim.x >> 6) + wave;
  if (s >= p.S) return;
  const int lane = threadIdx.x & 63;
  const int F = p.F, R = p.n + 2;
  float* rt_smem = rt_smem_all + (size_t)wave * p.lds_floats_per_wave;
  const long long n = (long long)R * F;
  const float* spec = p.spec + (long long)s * n;
  const float* tgw = p.tgradw + (long long)s * n;
  const float* fgw = p.fgradw + (long long)s * n;
  float* phase = p.phase + (long long)s * n;
  float* srow = rt_smem;       // working copy of spectrogram row f
  float* hrow = srow + F;      // untouched row f-1 (spectrogram_history, dgt.py:411)
  float* ph0 = hrow + F;       // phase row f-1
  float* ph1 = ph0 + F;        // phase row f
  float* tg0 = ph1 + F;        // padded tgradw rows f-1 and f
  float* tg1 = tg0 + F;
  float* fg1 = tg1 + F;        // padded fgradw row f
  const HeapT<false> H = {reinterpret_cast<u64*>(fg1 + F + (F & 1)), nullptr, 0x7fffffff};
  const u64 anc_mask = chain_mask(lane);
  auto lds_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto ufloat = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };

  float smax = -1.0f;
  for (long long i = lane; i < n; i += 64) smax = fmaxf(smax, spec[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) smax = fmaxf(smax, __shfl_xor(smax, o, 64));
  float abstol = p.tol * smax;  // :400
  if (abstol < p.eps) abstol = p.eps;

  // :402-403 row 1 of the phase array is the previous call's last row.  The phase row of the frame before and its time
  // gradient stay in LDS from frame to frame (round 5): the work array in global memory was written, fenced at agent
  // scope and read back by the same wave every frame -- a third of the scan path's time per frame.
  for (int k = lane; k < F; k += 64) {
    ph1[k] = p.prev_phase[(long long)s * F + k];
    tg1[k] = 0.0f;                       // :408-410 two-row front padding: the gradient row "before" frame 2
  }
  (void)phase;
  lds_sync();

  for (int f = 2; f < R; ++f) {  // :413
#ifdef AT_DEV_SWITCHES
    const unsigned long long tickf0 = wall_clock64();
#endif
    float max_val = -1.0f;
    long long max_k = F;
    for (int k = lane; k < F; k += 64) {   // rows f-1 of the phase and of tgradw: last frame's ph1 / tg1
      ph0[k] = ph1[k];
      tg0[k] = tg1[k];
    }
    lds_sync();
    // four strides of 64 bins per trip, all 28 loads requested before the first is used: one wave per SIMD hides nothing,
    // and a trip per 64 bins was nine dependent round trips to global memory per frame (11 us of the scan path's 27)
    for (int kb = lane; kb < F; kb += 256) {
      float v_[4], h_[4], nz_[4], t1_[4], g1_[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = kb + 64 * u;
        const bool in = k < F;
        const long long kk = in ? k : 0;
        v_[u] = spec[(long long)f * F + kk];
        h_[u] = spec[(long long)(f - 1) * F + kk];
        nz_[u] = p.noise[(long long)s * p.n * F + (long long)(f - 2) * F + kk];
        t1_[u] = tgw[(long long)(f - 2) * F + kk];
        g1_[u] = fgw[(long long)(f - 2) * F + kk];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = kb + 64 * u;
        if (k < F) {
          const float v = v_[u];
          srow[k] = v;
          hrow[k] = h_[u];
          ph1[k] = (v > abstol) ? 0.0f : nz_[u];  // :404-405
          tg1[k] = t1_[u];
          fg1[k] = g1_[u];
          if (v > max_val) {
            max_val = v;
            max_k = k;
          }
        }
      }
    }
    wave_argmax(max_val, max_k);
    max_val = ufloat(max_val);
    lds_sync();
    // Rank fast path (see pghi_rt_rank_kernel): the frontier is a bitmap over the ranks of the frame's 2F candidates.
    //
    // Ties.  Among 1026 float32 magnitudes SOME two are equal in ~2 % of all frames -- with 256 streams in nearly every
    // step -- so a tie cannot simply send the frame to the heap.  Tied entries leave the queue back to back in an order
    // only the heap knows (its sift rules, utils/heapq.py:9-59).  Call a tied entry's pop together with everything that
    // pops because of it before the next tied entry gets its turn (entries it pushes whose keys are larger) its BLOCK.
    // A block's effect depends on the state only through the bins it looks at (its own phase, the availability of its
    // targets): T; it changes the state only by visiting bins: V (subset of T).  If for every two blocks of a tie group
    // V of one misses T of the other, every block does the same whatever order the group is taken in, and the result
    // is the heap's.  That is checked as the group runs (in rank order); a collision abandons the fast path and the
    // frame is redone on the heap.  (tools/fuzz_rt_ties.py: injected ties, bit for bit against the heap kernel.)
    bool fast_done = false;
    if (p.scan_path && max_val > abstol) {
#ifdef AT_DEV_SWITCHES
      const unsigned long long tick0 = wall_clock64();
#endif
      float* scan_scratch = fg1 + F + (F & 1);            // where the heap would be
      const int kmx = uni((int)max_k);
      if (F <= 64 * 9) fast_done = rt_scan_frame<9>(F, lane, abstol, kmx, srow, hrow, ph0, ph1, tg0, tg1, fg1, scan_scratch);
      else if (F <= 64 * 17) fast_done = rt_scan_frame<17>(F, lane, abstol, kmx, srow, hrow, ph0, ph1, tg0, tg1, fg1, scan_scratch);
      // (rows of more than 1088 bins do not fit this kernel's LDS budget anyway: n_fft 4096 runs on the global-memory kernels)
      lds_sync();
#ifdef AT_DEV_SWITCHES
      RT_STAT(5, wall_clock64() - tick0);      // 100 MHz ticks inside the scan path
#endif
    }
    bool try_fast = !fast_done && p.ranks != nullptr && max_val > abstol;
    if (try_fast) {
      // Pervasive ties: a held frame, a hop-periodic tone or a test signal makes row f equal row f-1, every
      // pop then lands in a tie group whose block visits its tied partner and the frame is redone on the heap anyway --
      // after paying for the tables, the records and the aborted flood.  The pre-pass has already marked the tied ranks:
      // when more than a quarter of the 2F candidates are tied the attempt is skipped.
      const unsigned short* grec0 = p.ranks + ((long long)s * p.n + (f - 2)) * p.rank_stride;
      int nt = (lane < (2 * F + 31) / 32) ? __popc(reinterpret_cast<const unsigned*>(grec0 + 4 * F)[lane]) : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) nt += __shfl_xor(nt, o, 64);
      if (2 * nt > F) try_fast = false;
    }
    if (try_fast) {
      const unsigned short* grec = p.ranks + ((long long)s * p.n + (f - 2)) * p.rank_stride;
      // LDS: per-rank records where the heap would be (ONE ds_read_b128 per pop), the rank tables and two bitmaps behind
      // them (the launcher sized the allocation for that).  A record is {phase, targets, step up, step down}:
      //   row f-1, bin k:  {ph0[k],            rank of (f, k)   | 0xffff << 16,        0.5 (tg0[k] + tg1[k]),  0}
      //   row f,   bin k:  {ph1[k] (written    rank of (f, k+1) | rank of (f, k-1) << 16, 0.5 (fg[k] + fg[k+1]), 0.5 (fg[k] + fg[k-1])}
      //                     when reached),
      // so that one pop is: target `up` gets phase + step up, target `down` gets phase - step down, whatever the row
      // (dgt.py:438-440, 447-449, 455-457); 0xffff: no such target (k + 1 = F; k - 1 <= 0: bin 0 is never reached
      // downward, :453).
      // (an offset in floats, not a rounded address: a pointer that has been through uintptr_t is a generic pointer and
      //  every access through it a flat_load / flat_store)
      int4* recs = reinterpret_cast<int4*>(rt_smem + ((7 * F + 3) & ~3));         // [2F], 16-byte aligned (the wave's LDS is)
      unsigned short* eor = reinterpret_cast<unsigned short*>(recs + 2 * F);     // entry at rank r (e < F: row f-1, bin e; else row f, bin e - F)
      unsigned short* roe = eor + 2 * F;                                          // rank of entry e
      unsigned* bm = reinterpret_cast<unsigned*>(roe + 2 * F);                    // [0,64): row f-1 live, [64,128): row f live, by rank
      {
        const unsigned* src = reinterpret_cast<const unsigned*>(grec);
        unsigned* dst = reinterpret_cast<unsigned*>(eor);
        for (int i = lane; i < 2 * F; i += 64) dst[i] = src[i];              // 4F u16 = 2F words
        bm[lane] = 0u;
        bm[64 + lane] = 0u;
      }
      const unsigned sp = (lane < (2 * F + 31) / 32) ? reinterpret_cast<const unsigned*>(grec + 4 * F)[lane] : 0u;   // same-as-previous, by rank
      // rank r is part of a tie: same as its predecessor or as its successor (whose bit may sit in the next lane's word)
      const unsigned tied_bits = sp | (sp >> 1) | ((unsigned)__shfl_down((int)sp, 1, 64) << 31);
      lds_sync();
      for (int k = lane; k < F; k += 64) {
        const unsigned ra = roe[k], rb = roe[F + k];
        if (hrow[k] > abstol) atomicOr(&bm[ra >> 5], 1u << (ra & 31));
        if (srow[k] > abstol) atomicOr(&bm[64 + (rb >> 5)], 1u << (rb & 31));
        recs[ra] = make_int4(__float_as_int(ph0[k]), (int)(rb | 0xffff0000u), __float_as_int(0.5f * (tg0[k] + tg1[k])), 0);
        const unsigned ru = (k + 1 < F) ? (unsigned)roe[F + k + 1] : 0xffffu;
        const unsigned rd = (k - 1 > 0) ? (unsigned)roe[F + k - 1] : 0xffffu;
        const float gk = fg1[k];
        recs[rb] = make_int4(__float_as_int(ph1[k]), (int)(ru | (rd << 16)), __float_as_int(0.5f * (gk + fg1[k + 1 < F ? k + 1 : k])),
                             __float_as_int(0.5f * (gk + fg1[k >= 1 ? k - 1 : 0])));
      }
      lds_sync();
      unsigned qbits = bm[lane];          // the frontier, by rank: lane i holds ranks 32 i .. 32 i + 31
      unsigned bav = bm[64 + lane];       // row-f entries that are live and not yet visited, by rank
      // rec_word[4 r] = phase of rank r as its bit pattern: written through the SAME type the records are read with (int4
      // words), so the compiler may not reorder the lane-0 store against the next pop's 16-byte load (a float*
      // alias of the int4 records relied on in-order DS issue only)
      int* rec_word = reinterpret_cast<int*>(recs);
      auto bit_of = [&](unsigned bits, int i) -> bool {
        return (((unsigned)__builtin_amdgcn_readlane((int)bits, i >> 5)) >> (i & 31)) & 1u;
      };
      auto first_of = [&](unsigned bits, int& r) -> bool {       // lowest set rank of a lane-distributed bitmap
        const u64 nz = __ballot(bits != 0u);
        if (nz == 0) return false;
        const int L = __builtin_ctzll(nz);
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)bits, L);
        r = 32 * L + __builtin_ctz(w);
        return true;
      };
      // tie-group state: cur_* = the running block, acc_* = the group's earlier blocks; T / V are sets of entries by rank
      unsigned cur_v = 0u, cur_t = 0u, acc_v = 0u, acc_t = 0u;
      // one pop of rank r (already off the frontier), its record in hand.  TRACK: inside a tie group, with the block
      // bookkeeping.  Returns, for the caller's prefetch, whether rank `watch` had its phase written.
      auto process = [&](int r, const int4 rec, int watch, auto track) -> bool {
        constexpr bool TRACK = decltype(track)::value;
        const unsigned w1 = (unsigned)uni(rec.y);
        const int ru = (int)(w1 & 0xffffu), rd = (int)(w1 >> 16);
        const float pk = __int_as_float(rec.x);
        // availability of the two targets, without branching on "is there one": the lane index is masked, the verdict is not
        const unsigned wu = (unsigned)__builtin_amdgcn_readlane((int)bav, (ru >> 5) & 63);
        const unsigned wd = (unsigned)__builtin_amdgcn_readlane((int)bav, (rd >> 5) & 63);
        const bool up = ru != 0xffff && ((wu >> (ru & 31)) & 1u);
        const bool dn = rd != 0xffff && ((wd >> (rd & 31)) & 1u);
        if (up && lane == 0) rec_word[4 * ru] = __float_as_int(pk + __int_as_float(rec.z));
        if (dn && lane == 0) rec_word[4 * rd] = __float_as_int(pk - __int_as_float(rec.w));
        const unsigned mu = up ? (1u << (ru & 31)) : 0u, md = dn ? (1u << (rd & 31)) : 0u;
        const unsigned add = ((lane == (ru >> 5)) ? mu : 0u) | ((lane == (rd >> 5)) ? md : 0u);
        qbits |= add;                       // the reached entries join the frontier ...
        bav &= ~add;                        // ... and are no longer unvisited
        if (TRACK) {
          cur_v |= add;
          cur_t |= add | ((lane == (r >> 5)) ? (1u << (r & 31)) : 0u) |
                   ((ru != 0xffff && lane == (ru >> 5)) ? (1u << (ru & 31)) : 0u) |
                   ((rd != 0xffff && lane == (rd >> 5)) ? (1u << (rd & 31)) : 0u);
        }
        return (up && ru == watch) || (dn && rd == watch);
      };
      auto close_block = [&]() -> bool {    // the running block against the group's earlier ones; then it joins them
        const bool clash = __ballot(((cur_v & acc_t) | (cur_t & acc_v)) != 0u) != 0;
        acc_v |= cur_v;
        acc_t |= cur_t;
        cur_v = 0u;
        cur_t = 0u;
        return !clash;
      };
      bool ok = true;
      int r = 0;
      if (first_of(bav, r)) {               // :427 the frame maximum seeds the frontier and is NOT marked visited
        if (lane == (r >> 5)) qbits |= 1u << (r & 31);
      }
      while (true) {
        // ---- the common case: pops whose magnitude is nobody else's
        // (Requesting the next pop's record ahead -- the frontier's next rank as it stands, read again when this pop reaches
        //  something larger -- was built and measured: 0.26 -> 0.39 ms per frame; the select between the two records and
        //  the extra find-first cost more than the round trip they hide.)
        bool tied = false;
        while (first_of(qbits, r)) {
          if (bit_of(tied_bits, r)) {
            tied = true;
            break;
          }
          if (lane == (r >> 5)) qbits &= ~(1u << (r & 31));
          process(r, recs[r], -1, std::integral_constant<bool, false>());
        }
        if (!tied) {                        // :461-465 the frontier ran dry: the largest unvisited bin of the row, marked
          if (!first_of(bav, r)) break;     // (the reference pushes one last entry at or below abstol and leaves: no effect)
          if (lane == (r >> 5)) {
            bav &= ~(1u << (r & 31));
            qbits |= 1u << (r & 31);        // popped at once by the next trip
          }
          continue;
        }
        // ---- a tie group: ranks [g_lo, g_hi] share one magnitude; rank r, still on the frontier, is its first member out
        int g_lo = r, g_hi = r;
        while (bit_of(sp, g_lo)) --g_lo;                                  // (bit 0 is never set)
        while (g_hi + 1 < 2 * F && bit_of(sp, g_hi + 1)) ++g_hi;
        cur_v = cur_t = acc_v = acc_t = 0u;
        bool finished = false;
        while (true) {
          if (!first_of(qbits, r)) {        // reseed inside the group's span: part of the running block
            if (!first_of(bav, r)) {
              finished = true;
              break;
            }
            if (lane == (r >> 5)) {
              bav &= ~(1u << (r & 31));
              qbits |= 1u << (r & 31);
              cur_v |= 1u << (r & 31);
              cur_t |= 1u << (r & 31);
            }
            continue;
          }
          if (r > g_hi) break;              // the group is over; rank r stays on the frontier for the common loop
          if (lane == (r >> 5)) qbits &= ~(1u << (r & 31));
          if (r >= g_lo && !close_block()) {       // the next tied entry: the running block is complete
            ok = false;
            break;
          }
          process(r, recs[r], -1, std::integral_constant<bool, true>());
        }
        if (ok && !close_block()) ok = false;
        if (!ok || finished) break;
      }
      fast_done = ok;
      lds_sync();
      if (ok) {                             // back to bin order
        for (int k = lane; k < F; k += 64) ph1[k] = __int_as_float(rec_word[4 * roe[F + k]]);
        lds_sync();
      }
    }
    if (fast_done) {
      // nothing left to do for this frame
    } else if (max_val > abstol) {  // :416-417
      int hn = 1;
      if (lane == 0) H.store(0, pack_item(-max_val, F + (int)max_k));   // :427 the seed is NOT marked visited
      lds_sync();
      auto push = [&](float key, int idx) {     // heappush (heapq.py:45-48)
        coop_siftdown(H, hn, pack_item(key, idx), lane);
        ++hn;
        lds_sync();
      };
      // :428-430 every live bin of row f-1, in bin order.  64 bins per LDS read, then a scalar walk over the live
      // ones (one LDS round trip per bin costs as much as a fifth of the push it decides about)
      for (int k0 = 0; k0 < F; k0 += 64) {
        const float mine = (k0 + lane < F) ? hrow[k0 + lane] : 0.0f;
        u64 live_bins = __ballot(mine > abstol);
        while (live_bins) {
          const int j = __builtin_ctzll(live_bins);
          live_bins &= live_bins - 1;
          const float hv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), j));
          push(-hv, k0 + j);
        }
      }
      while (max_val > abstol) {  // :433
        while (hn > 0) {
          // heappop (heapq.py:51-59)
          const u64 top63 = H.top[(lane >= 1 && lane - 1 < 4 * F + 8) ? lane - 1 : 0];   // the heap holds 4F + 8 entries
          const u64 last = H.load(hn - 1);
          hn -= 1;
          const int idx = uni(item_idx(hn == 0 ? last : readlane64(top63, 1)));
          const bool cur_row = idx >= F;
          const int k = cur_row ? idx - F : idx;
          // the magnitudes this pop may propagate to, requested before the heap repair (they come back first:
          // LDS returns in order) instead of one round trip each afterwards
          const float s_here = srow[k];
          const float s_up = srow[k + 1 < F ? k + 1 : k];
          const float s_dn = srow[k >= 1 ? k - 1 : 0];
          if (hn > 0) {
            u64 leaf_old = 0;
            const int leaf = coop_bubble(H, hn, lane, anc_mask, leaf_old, top63);
            if (leaf == 0 || !(item_key(last) < item_key(leaf_old))) {
              if (lane == 0) H.store(leaf, last);
            } else {
              coop_siftdown(H, leaf, last, lane);
            }
            lds_sync();
          }
          if (!cur_row) {  // :436-443 propagate in time
            const float sv = ufloat(s_here);
            if (sv > abstol) {
              if (lane == 0) {
                ph1[k] = ph0[k] + 0.5f * (tg0[k] + tg1[k]);
                srow[k] = abstol;
              }
              push(-sv, F + k);
            }
          } else {  // :444-460 propagate in frequency
            if (k + 1 < F) {
              const float sv = ufloat(s_up);
              if (sv > abstol) {
                if (lane == 0) {
                  ph1[k + 1] = ph1[k] + 0.5f * (fg1[k] + fg1[k + 1]);
                  srow[k + 1] = abstol;
                }
                push(-sv, F + k + 1);
              }
            }
            if (k - 1 > 0) {  // bin 0 is never reached downward (:453)
              const float sv = ufloat(s_dn);
              if (sv > abstol) {
                if (lane == 0) {
                  ph1[k - 1] = ph1[k] - 0.5f * (fg1[k] + fg1[k - 1]);
                  srow[k - 1] = abstol;
                }
                push(-sv, F + k - 1);
              }
            }
          }
        }
        // :461-465 reseed inside the frame: first index of the row maximum, lane-parallel
        float mv = -3.402823466e+38f;
        long long mk = F;
        for (int k = lane; k < F; k += 64) {
          const float v = srow[k];
          if (v > mv) {
            mv = v;
            mk = k;
          }
        }
        wave_argmax(mv, mk);
        max_val = ufloat(mv);
        const int mki = uni((int)mk);
        push(-max_val, F + mki);
        if (lane == 0) srow[mki] = abstol;
        lds_sync();
      }
    }
    lds_sync();
    for (int k = lane; k < F; k += 64) p.phase_out[(long long)s * p.n * F + (long long)(f - 2) * F + k] = ph1[k];
    lds_sync();
#ifdef AT_DEV_SWITCHES
    RT_STAT(6, wall_clock64() - tickf0);     // 100 MHz ticks per frame, everything included
#endif
  }
}

// x = mag * exp(i phase): refresh the PGHI history (|x[-2:]|, angle(x[-1]))   dgt.py:325-336
struct RtUpdParams {
  const float* mag;       // (S, n, F)
  const float* phase;     // (S, n, F)
  const float* hist_in;   // (S, 2, F)
  float* hist_out;        // (S, 2, F)
  float* phase_out;       // (S, F)
  int S, n, F;
};

__global__ __launch_bounds__(256) void rt_update_kernel(RtUpdParams p) {
  const long long total = (long long)p.S * p.F;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(i / p.F), k = (int)(i - (long long)s * p.F);
    const long long last = ((long long)s * p.n + (p.n - 1)) * p.F + k;
    float sn, cs;
    sincosf(p.phase[last], &sn, &cs);
    const float re = p.mag[last] * cs, im = p.mag[last] * sn;
    float prev;
    if (p.n > 1) {
      const long long l2 = last - p.F;
      float s2, c2;
      sincosf(p.phase[l2], &s2, &c2);
      prev = hypotf(p.mag[l2] * c2, p.mag[l2] * s2);
    } else {
      prev = p.hist_in[((long long)s * 2 + 1) * p.F + k];
    }
    p.hist_out[((long long)s * 2) * p.F + k] = prev;
    p.hist_out[((long long)s * 2 + 1) * p.F + k] = hypotf(re, im);
    p.phase_out[i] = fast_atan2f(im, re);
  }
}

static inline unsigned grid1d(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace at_hip

using namespace at_hip;

extern "C" {

int at_pghi_gradients(const float* mag, int64_t B, int T, int F, float gamma, int n_fft, int hop, float eps,
                      float* tgradw, float* fgradw, float* spec_or_null, void* stream) {
  if (B < 0 || T <= 0 || F <= 0 || n_fft <= 0 || hop <= 0) return AT_EINVAL;
  if (B == 0) return AT_OK;
  if (!mag || !tgradw || !fgradw) return AT_EINVAL;
  GradParams p = {mag, spec_or_null, tgradw, fgradw, (long long)B, T, F, n_fft, hop, gamma, eps};
  hipLaunchKernelGGL(pghi_grad_offline_kernel, dim3(grid1d((long long)B * T * F)), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

size_t at_pghi_offline_workspace_bytes(int64_t B, int T, int F);
}
static int pghi_integrate_launch(float* spec, const float* tg, const float* fg, int64_t B, int T, int F, float tol, float abstol,
                                 float* phase, at_hip::HeapItem* heap, int64_t* npops_or_null, int32_t* order_or_null,
                                 hipStream_t s);
extern "C" {
size_t at_pghi_offline_workspace_bytes(int64_t B, int T, int F) {
  const size_t n = (size_t)T * (size_t)F;
  // spec + tgradw + fgradw (fp32) + heap (8 B entries, n + 2)
  return (size_t)B * (3 * n * sizeof(float) + (n + 2) * sizeof(HeapItem)) + 256;
}

int at_pghi_offline(const float* mag, int64_t B, int T, int F, float gamma, int n_fft, int hop, float tol, float abstol,
                    float* phase, void* workspace, size_t workspace_bytes, int64_t* npops_or_null,
                    int32_t* order_or_null, void* stream) {
  if (B < 0 || T <= 0 || F <= 0 || n_fft <= 0 || hop <= 0) return AT_EINVAL;
  if (B == 0) return AT_OK;
  if (!mag || !phase) return AT_EINVAL;
  // heap positions are 32-bit and a bubble round addresses ((pos + 1) << 5) + 31; the frame / bin split of a bin
  // index goes through fp32: both hold up to 2^26 bins per clip (12 minutes of audio at n_fft 1024, hop 256)
  if ((long long)T * F > (1LL << 26) - 64) return AT_EUNSUPPORTED;
  if (!workspace || workspace_bytes < at_pghi_offline_workspace_bytes(B, T, F)) return AT_EWORKSPACE;
  const size_t n = (size_t)T * (size_t)F;
  float* spec = (float*)workspace;
  float* tg = spec + (size_t)B * n;
  float* fg = tg + (size_t)B * n;
  uintptr_t hp = ((uintptr_t)(fg + (size_t)B * n) + 15) & ~(uintptr_t)15;
  HeapItem* heap = (HeapItem*)hp;
  hipStream_t s = (hipStream_t)stream;
  GradParams g = {mag, spec, tg, fg, (long long)B, T, F, n_fft, hop, gamma, abstol};
  hipLaunchKernelGGL(pghi_grad_offline_kernel, dim3(grid1d((long long)B * T * F)), dim3(256), 0, s, g);
  return pghi_integrate_launch(spec, tg, fg, B, T, F, tol, abstol, phase, heap, npops_or_null, order_or_null, s);
}

int at_pghi_integrate(const float* mag, const float* tgradw, const float* fgradw, int64_t B, int T, int F, float tol,
                      float abstol, float* phase, void* workspace, size_t workspace_bytes, int64_t* npops_or_null,
                      int32_t* order_or_null, void* stream) {
  if (B < 0 || T <= 0 || F <= 0) return AT_EINVAL;
  if (B == 0) return AT_OK;
  if (!mag || !tgradw || !fgradw || !phase) return AT_EINVAL;
  if ((long long)T * F > (1LL << 26) - 64) return AT_EUNSUPPORTED;
  if (!workspace || workspace_bytes < at_pghi_offline_workspace_bytes(B, T, F)) return AT_EWORKSPACE;
  const size_t n = (size_t)T * (size_t)F;
  float* spec = (float*)workspace;                       // the integration marks visited bins in its own copy
  uintptr_t hp = ((uintptr_t)(spec + 3 * (size_t)B * n) + 15) & ~(uintptr_t)15;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(spec, mag, sizeof(float) * (size_t)B * n, hipMemcpyDeviceToDevice, s) != hipSuccess) return AT_ELAUNCH;
  return pghi_integrate_launch(spec, tgradw, fgradw, B, T, F, tol, abstol, phase, (HeapItem*)hp, npops_or_null, order_or_null, s);
}

}  // extern "C"

// the heap integration proper: spec (B, T, F) is consumed (visited bins are overwritten), the gradients are read only
static int pghi_integrate_launch(float* spec, const float* tg, const float* fg, int64_t B, int T, int F, float tol, float abstol,
                                 float* phase, at_hip::HeapItem* heap, int64_t* npops_or_null, int32_t* order_or_null,
                                 hipStream_t s) {
  using namespace at_hip;
  static const int prof = [] { const char* e = dev_env("ACIDS_PGHI_PROF"); return (e && e[0] == '1') ? 1 : 0; }();   // dev builds only
  // LDS share of the heap: as much as fits while every clip of the batch can still be resident (160 KB per CU)
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  const long long per_cu = (B + cus - 1) / cus;
  const int cap = per_cu <= 1 ? 16383 : per_cu <= 2 ? 8191 : per_cu <= 4 ? 4095 : per_cu <= 8 ? 2047 : per_cu <= 16 ? 1023 : 511;
  // segment maxima for the reseeds (cooperative kernel), behind the heap's top: as many as keep per_cu clips resident
  const int seg_cap = per_cu <= 4 ? 1024 : per_cu <= 8 ? 512 : 192;
  const size_t heap_lds = sizeof(u64) * ((size_t)(cap + 1) + (size_t)(seg_cap + 1) / 2);
  HgiParams h = {spec, tg, fg, phase, heap, (long long)B, T, F, abstol, tol, (long long*)npops_or_null, cap, seg_cap, prof,
                 order_or_null};
  // at_set_variant(AT_VARIANT_PGHI_KERNEL, 2) selects the single-lane reference kernel (debugging aid; identical results),
  // 1 the winner-bit variant (identical results; slower on every batch measured, see the comment above it and DESIGN.md
  // 3.4 -- kept selectable so that the parity tests and tools/fuzz_pghi.py can run it)
  const int pghi_kernel = variant(kVarPghiKernel);
  const bool serial = pghi_kernel == 2;
  const bool use_coop = pghi_kernel != 1;
  if (serial) {
    hipLaunchKernelGGL(pghi_hgi_offline_kernel, dim3((unsigned)B), dim3(64), 0, s, h);
  } else if (!use_coop && !prof) {
    // winner-bit kernel: per clip 16.5 KB of bits + the heap's top levels in LDS; at most 8 clips resident per CU
    const int wcap = per_cu <= 4 ? 2047 : per_cu <= 6 ? 1023 : 255;
    h.heap_lds_cap = wcap;
    const size_t per_wave = sizeof(u64) * ((size_t)(wcap + 1) + (WB_WORDS + 1) / 2);
    int wpb = per_cu >= 8 ? 8 : per_cu >= 4 ? 4 : per_cu >= 2 ? 2 : 1;
    while (wpb > 1 && per_wave * wpb > 160 * 1024 - 512) wpb >>= 1;
    const size_t block_lds = per_wave * wpb;
    if (block_lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)pghi_hgi_offline_wbit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)block_lds) != hipSuccess) {
      (void)hipGetLastError();
      return AT_ELAUNCH;
    }
    hipLaunchKernelGGL(pghi_hgi_offline_wbit_kernel, dim3((unsigned)((B + wpb - 1) / wpb)), dim3(64 * wpb), block_lds, s, h);
  } else {
    void (*kernel)(HgiParams) = prof ? pghi_hgi_offline_coop_kernel<true> : pghi_hgi_offline_coop_kernel<false>;
    if (const char* e = dev_env("ACIDS_PGHI_PUSH"))       // dev builds: "batch" = the parent pre-test of the pushes (slower)
      if (!strcmp(e, "batch")) kernel = prof ? pghi_hgi_offline_coop_kernel<true, true> : pghi_hgi_offline_coop_kernel<false, true>;
    // waves per workgroup: as many (<= 8) as keep the workgroup's heap tops within the CU's 160 KB
    int wpb = per_cu >= 8 ? 8 : per_cu >= 4 ? 4 : per_cu >= 2 ? 2 : 1;
    while (wpb > 1 && heap_lds * wpb > 160 * 1024 - 1024) wpb >>= 1;
    const size_t block_lds = heap_lds * wpb;
    if (block_lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)block_lds) != hipSuccess) {
      (void)hipGetLastError();
      return AT_ELAUNCH;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)((B + wpb - 1) / wpb)), dim3(64 * wpb), block_lds, s, h);
  }
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

extern "C" {

size_t at_pghi_rt_workspace_bytes(int S, int n, int F) {
  const size_t per = (size_t)(n + 2) * (size_t)F;
  // spec, hist, tgradw, fgradw, phase (n + 2 rows each) + device-drawn noise (n rows) + heap + the rank records
  return (size_t)S * (6 * per * sizeof(float) + (4 * (size_t)F + 8) * sizeof(HeapItem) +
                      (size_t)n * (size_t)at_hip::rt_rank_stride_u16(F) * 2) + 512;
}

static int pghi_realtime_impl(const float* mag_hist, const float* mag, const float* prev_phase, const float* noise,
                              unsigned* rng_state, int S, int n, int F, float gamma, int n_fft, int hop, float tol, float eps,
                              float* phase, float* tgradw_or_null, float* fgradw_or_null, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (S < 0 || n <= 0 || F <= 0 || n_fft <= 0 || hop <= 0) return AT_EINVAL;
  if (S == 0) return AT_OK;
  if (!mag_hist || !mag || !prev_phase || !phase) return AT_EINVAL;
  if ((noise == nullptr) == (rng_state == nullptr)) return AT_EINVAL;     // exactly one source of draws
  if (!workspace || workspace_bytes < at_pghi_rt_workspace_bytes(S, n, F)) return AT_EWORKSPACE;
  const size_t per = (size_t)(n + 2) * (size_t)F;
  float* w = (float*)workspace;
  RtParams p;
  p.mag_hist = mag_hist; p.mag = mag; p.prev_phase = prev_phase; p.phase_out = phase;
  p.noise_gen = noise ? nullptr : w + 5 * (size_t)S * per;
  p.noise = noise ? noise : p.noise_gen;
  p.rng_state = rng_state;
  p.spec = w; p.hist = w + (size_t)S * per;
  p.tgradw = tgradw_or_null ? tgradw_or_null : w + 2 * (size_t)S * per;
  p.fgradw = fgradw_or_null ? fgradw_or_null : w + 3 * (size_t)S * per;
  p.phase = w + 4 * (size_t)S * per;
  uintptr_t hp = ((uintptr_t)(w + 6 * (size_t)S * per) + 15) & ~(uintptr_t)15;
  p.heap = (HeapItem*)hp;
  p.S = S; p.n = n; p.F = F; p.n_fft = n_fft; p.hop = hop; p.gamma = gamma; p.tol = tol; p.eps = eps;
  p.ranks = nullptr;
  p.rank_stride = rt_rank_stride_u16(F);
  p.scan_path = 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pghi_grad_rt_kernel, dim3(grid1d((long long)S * per)), dim3(256), 0, s, p);
  const size_t lds = sizeof(float) * (7 * (size_t)F + 1) + sizeof(HeapItem) * (4 * (size_t)F + 8);
  // at_set_variant(AT_VARIANT_PGHI_KERNEL, 2) selects the single-lane kernels (debugging aid; identical results), 3 the
  // cooperative heap kernel without the rank fast path
  const int pghi_kernel = variant(kVarPghiKernel);
  const bool serial_rt = pghi_kernel == 2;
  p.scan_path = pghi_kernel == 0;            // 4: the rank fast path without the scan path in front of it
  // the fast path keeps 2F 16-byte records where the heap would be and 8F + 512 bytes of tables behind them
  const size_t lds_fast = sizeof(float) * (7 * (size_t)F + 4) + 16 * (2 * (size_t)F) + 8 * (size_t)F + 512 + 16;
  // Default (variant 0): the scan path resolves a frame without a queue and declines only competing ties and islands of
  // several unreached bins (dense noise: none in 8e4 frames; sparse frames that decline are cheap on the heap), so the
  // rank pre-pass -- 84 us per call at 256 streams x 4 frames, more than the scan path's whole flood -- is not run:
  // scan -> heap.  Variant 4 keeps round 4's chain rank bitmap -> heap (and its tests), variant 3 the heap alone.
  bool rank_path = false;
  if (pghi_kernel == 4 && lds_fast <= 64 * 1024 && 2 * F <= 2048 && F >= 8 && (long long)S * n < (1LL << 31)) {
    rank_path = true;
    // rank pre-pass: one workgroup per (stream, new frame) sorts the frame's 2F candidates
    uintptr_t rp = ((uintptr_t)(p.heap + (size_t)S * (4 * (size_t)F + 8)) + 15) & ~(uintptr_t)15;
    p.ranks = (unsigned short*)rp;
    int n2 = 2;
    while (n2 < 2 * F) n2 <<= 1;
    hipLaunchKernelGGL(pghi_rt_rank_kernel, dim3((unsigned)((long long)S * n)), dim3(256), (size_t)n2 * sizeof(u64), s, p, n2);
  }
  if (lds <= 64 * 1024 && !serial_rt) {
    const size_t per_wave = ((rank_path ? (lds_fast > lds ? lds_fast : lds) : lds) + 15) & ~(size_t)15;
    // up to one stream per CU spreads best (256 streams: 3.5 ms alone on their CUs, 3.7 ms packed four to a CU);
    // from four per CU on, a workgroup of four loads the CU's SIMDs evenly (1024 streams: 4.15 -> 3.88 ms)
    int wpb = S >= 4 * 256 ? 4 : 1;
    while (wpb > 1 && per_wave * wpb > 160 * 1024 - 1024) wpb >>= 1;
    p.lds_floats_per_wave = (int)(per_wave / sizeof(float));
    const size_t block_lds = per_wave * wpb;
    if (block_lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)pghi_hgi_rt_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)block_lds) != hipSuccess) {
      (void)hipGetLastError();
      return AT_ELAUNCH;
    }
    hipLaunchKernelGGL(pghi_hgi_rt_coop_kernel, dim3((unsigned)((S + wpb - 1) / wpb)), dim3(64 * wpb), block_lds, s, p);
  }
  else if (lds <= 64 * 1024)
    hipLaunchKernelGGL(pghi_hgi_rt_lds_kernel, dim3((unsigned)S), dim3(64), lds, s, p);
  else
    hipLaunchKernelGGL(pghi_hgi_rt_kernel, dim3((unsigned)S), dim3(64), 0, s, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

int at_pghi_realtime(const float* mag_hist, const float* mag, const float* prev_phase, const float* noise, int S, int n,
                     int F, float gamma, int n_fft, int hop, float tol, float eps, float* phase, float* tgradw_or_null,
                     float* fgradw_or_null, void* workspace, size_t workspace_bytes, void* stream) {
  if (!noise) return AT_EINVAL;
  return pghi_realtime_impl(mag_hist, mag, prev_phase, noise, nullptr, S, n, F, gamma, n_fft, hop, tol, eps, phase,
                            tgradw_or_null, fgradw_or_null, workspace, workspace_bytes, stream);
}

int at_pghi_realtime_seeded(const float* mag_hist, const float* mag, const float* prev_phase, uint32_t* rng_state, int S, int n,
                            int F, float gamma, int n_fft, int hop, float tol, float eps, float* phase, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!rng_state) return AT_EINVAL;
  return pghi_realtime_impl(mag_hist, mag, prev_phase, nullptr, rng_state, S, n, F, gamma, n_fft, hop, tol, eps, phase, nullptr,
                            nullptr, workspace, workspace_bytes, stream);
}

int at_rt_update_buffers(const float* mag, const float* phase, int S, int n, int F, const float* hist_in,
                         float* hist_out, float* phase_out, void* stream) {
  if (S < 0 || n <= 0 || F <= 0) return AT_EINVAL;
  if (S == 0) return AT_OK;
  if (!mag || !phase || !hist_in || !hist_out || !phase_out) return AT_EINVAL;
  RtUpdParams p = {mag, phase, hist_in, hist_out, phase_out, S, n, F};
  hipLaunchKernelGGL(rt_update_kernel, dim3(grid1d((long long)S * F)), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? AT_OK : AT_ELAUNCH;
}

}  // extern "C"

#ifdef AT_DEV_SWITCHES
extern "C" int at_dev_rt_stats(unsigned* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(at_hip::g_rt_stats), 8 * sizeof(unsigned)) != hipSuccess) return -5;
  if (reset) {
    unsigned z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(at_hip::g_rt_stats), z, sizeof(z)) != hipSuccess) return -5;
  }
  return 0;
}
#endif
