#!/bin/bash
# (round 4: the switches exist only in the DEVELOPMENT build of the library -- tools/README.md -- load it with
#  ACIDS_HIP_LIB=tools/ab/libacids_dev.so)
# A/B of the small projection's forms on one box: tools/ab_env_dct.sh [rounds]
for i in $(seq 1 ${1:-2}); do
  for f in regs regs_nt lds lds_nt; do
    echo "== $f"; ACIDS_PROJECT_SMALL_FORM=$f PERF_N=40 PERF_WARM=25 timeout -k 10 120 python tools/perf_all.py dct40 || exit 1
  done
done
