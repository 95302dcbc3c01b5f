#!/bin/bash
# Round 4: persistent-workgroup forward (tiles of 8 short runs in address order) against the one-long-run-per-wave launch.
echo "== PW off"; ACIDS_FWD_PW=0 PERF_N=40 PERF_WARM=25 timeout -k 10 200 python tools/perf_all.py fwd,fused || exit 1
for g in ${PW_RUNS:-4 6 8 12 16 32}; do
  echo "== PW run $g"; ACIDS_FWD_PW=1 ACIDS_FWD_PW_RUN=$g PERF_N=40 PERF_WARM=25 timeout -k 10 200 python tools/perf_all.py fwd,fused || exit 1
done
echo "== PW off"; ACIDS_FWD_PW=0 PERF_N=40 PERF_WARM=25 timeout -k 10 200 python tools/perf_all.py fwd,fused || exit 1
