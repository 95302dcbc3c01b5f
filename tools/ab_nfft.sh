#!/bin/bash
# A/B of the other FFT sizes and the stand-alone projection: tools/ab/libacids_before.so against the in-tree library.
for i in 1 2; do
  echo "== before round $i"; ACIDS_HIP_LIB=$PWD/tools/ab/libacids_before.so python tools/nfft_probe.py 512,2048,4096 2>/dev/null; ACIDS_HIP_LIB=$PWD/tools/ab/libacids_before.so python tools/perf_all.py mel128,mel513 2>/dev/null
  echo "== in-tree round $i"; python tools/nfft_probe.py 512,2048,4096 2>/dev/null; python tools/perf_all.py mel128,mel513 2>/dev/null
done
