#!/bin/bash
# Run ON THE GPU BOX: kernel-only durations (rocprofv3 --kernel-trace) of one tools/perf_all.py invocation.
# usage: tools/ktime.sh <tag> <perf_all modes>      (environment such as ACIDS_HIP_LIB is inherited)
TAG=$1; MODES=$2
REPO=$(pwd); OUT=$REPO/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 $REPO/tools/perf_all.py $MODES > $OUT/out.log 2> $OUT/err.log
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, "**/*kernel_trace.csv"), recursive=True)[0]
d = defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v2 = sorted(v)[: max(1, len(v) * 3 // 4)]
    tail = v[len(v) // 2:]                      # launches in issue order: the second half runs on warm clocks and touched pages
    print("%-8s %-72s n=%3d avg %8.1f us  min %8.1f  (fastest 3/4 avg %8.1f, second half avg %8.1f)" % (tag, k, len(v), sum(v) / len(v), min(v), sum(v2) / len(v2), sum(tail) / len(tail)))
PY
find $OUT -name "*.csv" -delete
