#!/bin/bash
# Run ON THE GPU BOX: per-launch durations, in issue order, of the kernels whose name contains <pattern>.
# usage: tools/ktrace_seq.sh <tag> <pattern> <script.py> [args...]
TAG=$1; PAT=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/kts_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 $REPO/"$@" > $OUT/out.log 2> $OUT/err.log
cd $REPO
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, "**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")[:60]
    if pat in k:
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, "n=%d" % len(v))
    print("   ", " ".join("%.0f" % t for t in v))
PY
find $OUT -name "*.csv" -delete
