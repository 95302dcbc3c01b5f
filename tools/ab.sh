#!/bin/bash
# Same-box A/B of two builds of the library (tools/ab/libacids_a.so against the in-tree one), alternating runs.
#   tools/ab.sh [kernels] [rounds]      kernels: a tools/perf_all.py selection (default fwd,inv,fused,fusedfeat)
K=${1:-fwd,inv,fused,fusedfeat}
R=${2:-3}
A=${ACIDS_AB_A:-tools/ab/libacids_a.so}
for i in $(seq 1 $R); do
  echo "== A ($A) round $i"; ACIDS_HIP_LIB=$PWD/$A python tools/perf_all.py $K || exit 1
  echo "== B (in-tree) round $i"; python tools/perf_all.py $K || exit 1
done
