#!/bin/bash
# Run ON THE GPU BOX: one SQ counter pass over offline PGHI only (PGHI_B clips), prints per-pop figures.
# usage: tools/pmc_pghi_quick.sh <tag> [env assignments...]
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmcq_$TAG; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
export STREAM_STEPS=0 PGHI_REPS=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p1 -o pmc -- python3 $REPO/tools/perf_pghi_stream.py > $OUT/p1.log 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -o pmc -- python3 $REPO/tools/perf_pghi_stream.py > $OUT/p2.log 2> $OUT/p2.err
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, os, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")[:50]
        if "pghi_hgi" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
B = int(os.environ.get("PGHI_B", "1024"))
pops = B * 353468.0
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) * 4
    print(tag, k)
    print("  cycles/pop %.0f  wait %.0f (%.0f%%)  issue-stall %.0f  issuing %.0f" % (
        wc / pops, m.get("SQ_WAIT_ANY", 0) * 4 / pops, 100 * m.get("SQ_WAIT_ANY", 0) * 4 / max(wc, 1),
        m.get("SQ_WAIT_INST_ANY", 0) * 4 / pops, m.get("SQ_ACTIVE_INST_ANY", 0) * 4 / pops))
    print("  per pop: VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.1f VMEM_WR %.1f SMEM %.1f" % tuple(
        m.get(c, 0) / pops for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")))
    print("  active quad-cycles per pop: LDS %.0f VALU %.0f SCA %.0f VMEM %.0f" % tuple(
        m.get(c, 0) / pops for c in ("SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM")))
PY
find $OUT -name "*.csv" -size +1M -delete
