#!/bin/bash
# Same-box comparison of the in-tree library with any number of variant builds, alternating runs.
#   tools/ab3.sh "<lib1> <lib2> ..." [kernels] [rounds]
LIBS=$1; K=${2:-fwd,inv,fused,fusedfeat}; R=${3:-3}
for i in $(seq 1 $R); do
  echo "== in-tree round $i"; python tools/perf_all.py $K 2>/dev/null || exit 1
  for L in $LIBS; do echo "== $L round $i"; ACIDS_HIP_LIB=$PWD/$L python tools/perf_all.py $K 2>/dev/null || exit 1; done
done
