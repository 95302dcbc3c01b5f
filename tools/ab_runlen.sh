#!/bin/bash
# Run length of the n_fft-1024 forward / fused forward (ACIDS_FWD_FPR) and inverse (ACIDS_ISTFT_SPR) kernels, one box.
for f in 0 87 58 44 35 29 22; do
  echo "== frames per run $f"
  if [ $f = 0 ]; then PERF_N=40 PERF_WARM=25 timeout -k 10 200 python tools/perf_all.py fwd,fused,inv || exit 1
  else ACIDS_FWD_FPR=$f ACIDS_ISTFT_SPR=$f PERF_N=40 PERF_WARM=25 timeout -k 10 200 python tools/perf_all.py fwd,fused,inv || exit 1; fi
done
