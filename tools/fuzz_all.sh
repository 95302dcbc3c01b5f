#!/bin/bash
# Every fuzz tool once with fresh seeds against the in-tree library (run ON THE GPU BOX): tools/fuzz_all.sh <seed base> <out file>
S=${1:-50}; OUT=${2:-gpurun_out/fuzz_all.txt}
: > $OUT
i=0
for t in fuzz_sizes fuzz_stft fuzz_banded fuzz_pghi fuzz_rtpghi fuzz_rt_ties fuzz_scans fuzz_polar fuzz_mulaw; do
  i=$((i+1))
  echo "== $t seed $((S+i))" >> $OUT
  FUZZ_SEED=$((S+i)) timeout -k 10 600 python tools/$t.py 2>&1 | tail -3 >> $OUT
  echo "rc=$?" >> $OUT
done
echo "== big_batch_check" >> $OUT
timeout -k 10 600 python tools/big_batch_check.py 2>&1 | tail -4 >> $OUT
tail -60 $OUT
