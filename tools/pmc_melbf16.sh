#!/bin/bash
# Run ON THE GPU BOX: MFMA / VALU busy counters and HBM traffic of the bf16 projection at 1024 x 690 frames.
TAG=${1:-r02}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_melbf16_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/p1 -o pmc -- python3 $REPO/tools/perf_all.py melbf16 > $OUT/p1.log 2> $OUT/p1.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o pmc -- python3 $REPO/tools/perf_all.py melbf16 > $OUT/p2.log 2> $OUT/p2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -o pmc -- python3 $REPO/tools/perf_all.py melbf16 > $OUT/p3.log 2> $OUT/p3.err
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")[:50]
        if "mel_bf16_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, {c: "%.4g" % x for c, x in sorted(m.items())})
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum over the 1024 SIMDs
        print("   MFMA pipe busy: %.2f %% of SIMD-cycles" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
print(open(os.path.join(out, "p1.log")).read())
PY
find $OUT -name "*.csv" -size +1M -delete
