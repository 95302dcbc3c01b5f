#!/bin/bash
# Round 4: SHORT runs (8 ... 32 frames per wave) of the n_fft-1024 forward / fused forward / inverse kernels, one box.
# tools/ubench/stream_pattern3.hip says a dispatch-ordered launch of short runs is the fastest shape for the memory system.
for f in 0 8 12 16 24 32 0; do
  echo "== frames per run $f"
  if [ $f = 0 ]; then PERF_N=40 PERF_WARM=25 timeout -k 10 200 python tools/perf_all.py fwd,fused,inv || exit 1
  else ACIDS_FWD_FPR=$f ACIDS_ISTFT_SPR=$f PERF_N=40 PERF_WARM=25 timeout -k 10 200 python tools/perf_all.py fwd,fused,inv || exit 1; fi
done
