// variants.h -- how a launcher chooses between kernels that compute the SAME result.
//
// Product builds decide from the call's arguments alone, plus the explicit variant table below, which the C ABI exposes
// as at_set_variant / at_get_variant (include/acids_hip.h): the parity tests use it to run the generic forms of kernels
// whose headline shapes have a specialised form, and compare the two.  One relaxed atomic load on the launch path.
//
// The environment switches of the kernel A/B scripts (tools/ab*.sh) exist only in -DAT_DEV_SWITCHES builds
// (make EXTRA=-DAT_DEV_SWITCHES): dev_env() is getenv() there and a constant nullptr in the product library, so the
// compiler drops the branches behind it.
#pragma once
#include <stdlib.h>

namespace at_hip {

enum {
  kVarEpilogue = 0,         // 0: fixed-length epilogue / projection where the bank has the headline shape; 1: always generic
  kVarFrameKernels = 1,     // 1: frame-at-a-time forward at n_fft 512 / 2048 / 4096 instead of the sliding-window kernels
  kVarSmallProjection = 2,  // 0: matrix-core form of the K <= 128 projection (the DCT behind MFCC); 1: row kernel
  kVarScanLayout = 3,       // 0: one block per clip for rows that are not whole 64-byte segments; 1: flattened columns
  kVarPghiKernel = 4,       // 0: cooperative heap kernels (+ rank fast path, realtime); 1: winner-bit offline kernel;
                            // 2: single-lane kernels; 3: cooperative kernels, realtime on the heap only; 4: realtime with the
                            // rank fast path but without the wavefront-parallel scan path in front of it
  kVarIstftRuns = 5,        // 1: the n_fft-1024 inverse always as one long run per wave (no workgroup tiles with LDS hand-over)
  kVarCount = 6
};

int variant(int which);     // capi.hip

#ifdef AT_DEV_SWITCHES
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

}  // namespace at_hip
