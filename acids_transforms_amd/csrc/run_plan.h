// run_plan.h -- how the streaming STFT kernels cut clips into per-wave runs (shared by the size-specific launchers).
#pragma once
#include <hip/hip_runtime.h>

namespace at_hip {

inline int num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// wave slots the chip offers one kernel variant (occupancy x CUs), cached per variant by the caller
template <typename K>
inline long long resident_waves(K kernel, int block_threads, size_t dyn_lds) {
  struct Entry { const void* k; size_t lds; long long waves; };
  static thread_local Entry cache[16];
  static thread_local int n_cached = 0;
  for (int i = 0; i < n_cached; ++i)
    if (cache[i].k == (const void*)kernel && cache[i].lds == dyn_lds) return cache[i].waves;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block_threads, dyn_lds) != hipSuccess || nb <= 0) nb = 2;
  const long long waves = (long long)nb * (block_threads / 64) * num_cus();
  if (n_cached < 16) cache[n_cached++] = {(const void*)kernel, dyn_lds, waves};
  return waves;
}

// Split each of the B clips' `units` (frames / hop slots) into equal runs, one wave per run.  Every wave of a
// streaming kernel takes the same time, so the launch costs ceil(waves / slots) rounds of (run length +
// per-run overhead): 4096 waves on 3072 slots is two rounds, the second one a third full -- pick the run
// count that minimises rounds x run length instead of aiming at a fixed wave count.
inline long long plan_units_per_run(long long B, long long units, long long slots, long long min_units,
                                    long long overhead_units) {
  long long best_upr = units, best_cost = -1;
  long long max_runs = units / (min_units > 0 ? min_units : 1);
  if (max_runs < 1) max_runs = 1;
  for (long long runs = 1; runs <= max_runs; ++runs) {
    const long long upr = (units + runs - 1) / runs;
    const long long waves = B * ((units + upr - 1) / upr);
    const long long cost = ((waves + slots - 1) / slots) * (upr + overhead_units);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best_upr = upr;
    }
  }
  return best_upr;
}

// The plain forwards (no epilogue) with work for every slot at least twice over take SHORT runs instead of the planner's
// one long run per wave: a workgroup's runs are then one tile of the output stream and the hardware dispatches the tiles
// in address order -- the compact, advancing write front of profiles/r04_launch_shape.md (n_fft 1024: 0.730 -> 0.699 ms).
// Only there: at n_fft 2048 / 4096 (8-frame runs) the same cut is 5-10 % SLOWER (a run start re-reads 3/4 of a longer
// window), at 512 it is lost in the noise.
inline long long short_runs_if_full(long long B, long long units, long long slots, long long planned, long long target) {
  if (planned > target && B * ((units + target - 1) / target) >= 2 * slots) return target;
  return planned;
}

}  // namespace at_hip
