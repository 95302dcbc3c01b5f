#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the kernels outside the bench step (tools/perf_all.py modes),
# condensed into gpurun_out/secondary_<tag>.md
TAG=${1:-r01}
REPO=$(pwd); OUT=$REPO/gpurun_out/sec_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sec -- python3 $REPO/tools/perf_all.py fwd,inv,polar,mel128,melbf16,mel513,fused513,fused2,mfcc40,dct40,phase,polarfwd,stftpolar,sinebank,sizes > $OUT/perf_all.log 2> $OUT/err.log
cd $REPO
python3 - "$OUT" "$TAG" > gpurun_out/secondary_$TAG.md <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(out, "**/*kernel_trace.csv"), recursive=True)[0]
d = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")
    d[n[:80]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 kernel trace of the kernels outside the bench step (`%s`)\n" % tag)
print("Command: `rocprofv3 --kernel-trace --stats -- python3 tools/perf_all.py fwd,inv,polar,mel128,melbf16,mel513,fused513,fused2,"
      "mfcc40,phase,polarfwd,stftpolar,sinebank,sizes` at 1024 clips x 4 s (690 frames x 513 bins at n_fft 1024; `sizes`: n_fft 4096 / "
      "2048 / 512 / 400), one MI355X.  `steady` = mean of "
      "the fastest three quarters of the calls (the first launches of a process run on cold clocks).\n")
print("| kernel | calls | avg us | steady us | min us |")
print("|---|---|---|---|---|")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("at::") or "rocclr" in k or sum(v) < 50:
        continue
    v2 = sorted(v)[: max(1, len(v) * 3 // 4)]
    print("| `%s` | %d | %.1f | %.1f | %.1f |" % (k, len(v), sum(v) / len(v), sum(v2) / len(v2), min(v)))
print("\n## tools/perf_all.py output of the same run (HIP-event timings incl. launch overhead)\n\n```")
for line in open(os.path.join(out, "perf_all.log")):
    if " ms" in line or "TFLOP" in line:
        print(line.rstrip())
print("```")
PY
find $OUT -name "*.csv" -delete
