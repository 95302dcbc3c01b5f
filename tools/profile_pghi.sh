#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 evidence for the PGHI and streaming kernels.
#   pass 1: --kernel-trace --stats               (durations)
#   pass 2/3: --pmc SQ counters, own runs (no trace domains next to --pmc)
# usage: tools/profile_pghi.sh <tag>
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pghi_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/perf_pghi_stream.py > $OUT/trace.log 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -o pmc -- python3 $REPO/tools/perf_pghi_stream.py > $OUT/pmc1.log 2> $OUT/pmc1.err
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -o pmc -- python3 $REPO/tools/perf_pghi_stream.py > $OUT/pmc2.log 2> $OUT/pmc2.err
echo "pmc2 rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc -- python3 $REPO/tools/perf_pghi_stream.py > $OUT/pmc3.log 2> $OUT/pmc3.err
echo "pmc3 rc=$?"
cd $REPO
python3 tools/summarize_pghi_profile.py $OUT $TAG > $OUT/summary_$TAG.md 2> $OUT/summarize.err
echo "summary rc=$?"
find $OUT -name "*.csv" -size +8M -delete
tail -40 $OUT/summary_$TAG.md
