#!/bin/bash
# Run ON THE GPU BOX: per-kernel durations (rocprofv3 --kernel-trace) of an arbitrary python script.
# usage: tools/ktime_any.sh <tag> <script.py> [args...]
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 $REPO/"$@" > $OUT/out.log 2> $OUT/err.log
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, "**/*kernel_trace.csv"), recursive=True)[0]
d = defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-8s %-60s n=%4d avg %9.1f us  total %8.2f ms  %5.1f%%" % (tag, k, len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / tot))
PY
find $OUT -name "*.csv" -delete
