#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <tag> "<counters>" <python args...>
TAG=$1; CNT=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -o pmc -- python3 $REPO/tools/perf_all.py "$@" > $OUT/run.log 2> $OUT/run.err
cd $REPO
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("at_hip::","")[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "at::native" in k or "rocclr" in k: continue
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "n=%d"%len(next(iter(v.values()))))
PY
