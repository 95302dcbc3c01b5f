#!/bin/bash
# (round 4: the switches exist only in the DEVELOPMENT build of the library -- tools/README.md -- load it with
#  ACIDS_HIP_LIB=tools/ab/libacids_dev.so)
# Same-box A/B over an environment switch of the in-tree library, alternating runs.
#   tools/ab_env.sh VAR "val1 val2 ..." [kernels] [rounds]
VAR=$1; VALS=$2; K=${3:-fwd,fused}; R=${4:-3}
for i in $(seq 1 $R); do
  for v in $VALS; do
    echo "== $VAR=$v round $i"; env $VAR=$v python tools/perf_all.py $K 2>/dev/null || exit 1
  done
done
