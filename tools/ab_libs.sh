#!/bin/bash
# Same-box alternating A/B over BUILDS of the library (run ON THE GPU BOX via gpurun).
#   tools/ab_libs.sh "<so> <so> ..." <perf_all modes> [rounds] [grep pattern]
# "0" stands for the in-tree library; other names are files under tools/ab/ (make -C acids_transforms_amd/csrc
# BUILD=build_x OUT=../../tools/ab/libacids_x.so EXTRA=-D...).  Settled clocks: PERF_WARM=20 PERF_N=40.
LIBS=$1; MODES=$2; R=${3:-3}; PAT=${4:-ms}
export PERF_WARM=${PERF_WARM:-20} PERF_N=${PERF_N:-40}
python tools/memprobe.py 2>/dev/null
for i in $(seq 1 $R); do
  for x in $LIBS; do
    if [ "$x" = 0 ]; then L=acids_transforms_amd/libacids_hip.so; else L=tools/ab/$x; fi
    echo "== $x round $i"
    ACIDS_HIP_LIB=$PWD/$L python tools/perf_all.py $MODES 2>/dev/null | grep -i "$PAT" | cut -c1-64 | tr -s ' ' | tr '\n' '|'; echo
  done
done
