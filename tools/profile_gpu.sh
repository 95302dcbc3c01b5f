#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats, then PMC passes in their own runs.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 20 --warmup 5 --settle-steps 60 --no-cpu-baseline --no-extras}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo "write rc=$?"
cd $REPO
python3 tools/summarize_profile.py $OUT $TAG > $OUT/summary_$TAG.md 2> $OUT/summarize.err
echo "summary rc=$?"
# keep only small artefacts for merge-back
find $OUT -name "*.csv" -size +8M -delete
ls -la $OUT $OUT/* | head -40
