#!/bin/bash
# kernel mix of the streaming step (eager) under rocprofv3
REPO=$(pwd); OUT=$REPO/gpurun_out/r05r_stream_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STREAMS=256 CHUNK=${CHUNK:-1024} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/tools/stream_prof.py > $OUT/log.txt 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-72s calls %4d  avg %8.1f us  %5.1f %%" % (k, len(v), sum(v) / len(v), 100 * sum(v) / tot))
PY
