"""Condense rocprofv3 CSV output (kernel trace + separate PMC passes) into a small markdown/JSON summary."""
import csv
import glob
import json
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
    return name[:90]


print("# rocprofv3 summary `%s`\n" % tag)
STEPS, WARM = int(os.environ.get("PROFILE_STEPS", "20")), int(os.environ.get("PROFILE_WARMUP", "5"))
SETTLE = int(os.environ.get("PROFILE_SETTLE", "60"))          # bench.py --settle-steps (untimed, before the warm-up steps)
print("Command: `python3 bench.py --steps %d --warmup %d --no-cpu-baseline --no-extras` (B=1024 clips x 4 s, 1 GPU; "
      "%d settle steps before the warm-up)\n" % (STEPS, WARM, SETTLE))
WARM += SETTLE
print("`timed avg` = the launches of bench.py's timed region only (the last %d of a step kernel before the side "
      "measurements): the first launches run on cold clocks and untouched pages and are what `avg` over all calls "
      "adds to it.\n" % STEPS)
kt = find("trace/**/*kernel_trace.csv")
dur = defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in dur.values())
    print("## kernel trace (--kernel-trace --stats)\n")
    print("| kernel | calls | avg us | timed avg us | min us | max us | total ms | % |")
    print("|---|---|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
        timed = v[WARM:WARM + STEPS] if len(v) >= WARM + STEPS else v
        print("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.3f | %.1f |" % (
            k, len(v), sum(v) / len(v), sum(timed) / len(timed), min(v), max(v), sum(v) / 1e3, 100 * sum(v) / total))
    print()
traffic = {}
for name, patt, col in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv", "fetch"),
                        ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv", "write")):
    f = find(patt)
    if not f:
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == name:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        traffic.setdefault(k, {})[col] = sum(v) / len(v)
if traffic:
    print("## HBM traffic per launch (separate --pmc passes; FETCH_SIZE/WRITE_SIZE are in KiB)\n")
    print("FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced streams (MI355X_MICROARCH.md, HBM):")
    print("the corrected read column doubles it.  Access widths here are 8 B/lane (512 B per wave instruction), so the")
    print("correction is applied as the guide prescribes and flagged as uncalibrated for this width.\n")
    print("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | read MB (x2 corrected) | write MB | total MB |")
    print("|---|---|---|---|---|---|")
    js = {}
    for k, v in sorted(traffic.items(), key=lambda kv: -(kv[1].get("fetch", 0) + kv[1].get("write", 0)))[:10]:
        fe, wr = v.get("fetch", 0.0), v.get("write", 0.0)
        rd_mb, wr_mb = 2 * fe * 1024 / 1e6, wr * 1024 / 1e6
        print("| `%s` | %.0f | %.0f | %.1f | %.1f | %.1f |" % (k, fe, wr, rd_mb, wr_mb, rd_mb + wr_mb))
        js[k] = {"fetch_kib": fe, "write_kib": wr, "read_bytes_corrected": 2 * fe * 1024, "write_bytes": wr * 1024}
    json.dump(js, open(os.path.join(out, "pmc_traffic_%s.json" % tag), "w"), indent=1)
    # what bench.py reports as roofline.traffic: per-launch HBM bytes of the step kernels from THIS pass
    bt = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `%s`; FETCH_SIZE "
                      "doubled per MI355X_MICROARCH.md (gfx950)" % tag}
    for key, needle in (("stft_fwd", "stft1024_h256_fwd_kernel<false, 1, 8, true, 0, false, 2, 2, 8, 2"), ("stft_fwd_unfused", "stft1024_h256_fwd_kernel<false, 0"),
                        ("istft", "istft1024_tile_kernel"), ("istft", "istft1024_ola_kernel"), ("mel", "mel_banded_kernel")):
        if key in bt:
            continue
        for k, v in js.items():
            if needle in k:
                bt[key] = int(round(v["read_bytes_corrected"] + v["write_bytes"], -5))
                break
    json.dump(bt, open(os.path.join(out, "bench_traffic_%s.json" % tag), "w"), indent=1)
