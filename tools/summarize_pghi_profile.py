"""Condense tools/profile_pghi.sh's rocprofv3 CSVs (kernel trace + SQ counter passes) into markdown."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.split("(")[0]
    for pre in ("void ", "at_hip::"):
        name = name.replace(pre, "")
    return name[:80]


KEEP = ("pghi", "oadd", "stft", "irfft", "mel_bf16", "mag_", "rt_update", "rfft", "pointwise")
print("# rocprofv3 summary `%s`: PGHI and streaming kernels\n" % tag)
print("Command: `python3 tools/perf_pghi_stream.py` under `tools/profile_pghi.sh` -- offline PGHI on %s dense-noise clips x "
      "4 s (two calls), then 20 eager steps each of the 256-stream per-hop (256-sample) and 1024-sample streaming sessions "
      "(bf16 MFMA mel).\n" % os.environ.get("PGHI_B", "1024"))
kt = find("trace/**/*kernel_trace.csv")
if kt:
    dur = defaultdict(list)
    for r in csv.DictReader(open(kt)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in dur.values())
    print("## kernel trace (--kernel-trace --stats)\n")
    print("| kernel | calls | avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if not any(s in k for s in KEEP) or sum(v) / total < 1e-4:
            continue
        print("| `%s` | %d | %.1f | %.1f | %.1f | %.3f | %.2f |" % (k, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3,
                                                                 100 * sum(v) / total))
    print()
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "pghi_hgi" in k or "mel_bf16" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc:
    print("## SQ counters per launch (separate --pmc passes; SQ_*_CYCLES / WAIT / ACTIVE count quad-cycles summed over waves)\n")
    for k, v in acc.items():
        print("### `%s`\n" % k)
        print("| counter | mean per launch | launches |")
        print("|---|---|---|")
        for c in sorted(v):
            print("| %s | %.4g | %d |" % (c, sum(v[c]) / len(v[c]), len(v[c])))
        m = {c: sum(x) / len(x) for c, x in v.items()}
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            wc = m["SQ_WAVE_CYCLES"]
            print("\nShare of wave lifetime: waiting (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%; "
                  "VALU %.3g + SALU %.3g instructions per launch.\n"
                  % (100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc,
                     100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0)))
