#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the forward kernels (plain / fused with spectrum / features only) at 1024 x 4 s.
TAG=${1:-r02}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_fused_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p1 -o pmc -- python3 $REPO/tools/perf_all.py fwd,fused,fused2,inv > $OUT/p1.log 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p2 -o pmc -- python3 $REPO/tools/perf_all.py fwd,fused,fused2,inv > $OUT/p2.log 2> $OUT/p2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p3 -o pmc -- python3 $REPO/tools/perf_all.py fwd,fused,fused2,inv > $OUT/p3.log 2> $OUT/p3.err
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")[:64]
        if "stft1024" in k or "istft1024" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
frames = 1024 * 690.0
for k, v in sorted(acc.items()):
    m = {c: sorted(x)[len(x) // 2] for c, x in v.items()}      # median launch (small warm-up launches excluded)
    wc = m.get("SQ_WAVE_CYCLES", 0) * 4
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "launches", len(v["SQ_WAVES"]), "waves %d" % m.get("SQ_WAVES", 0))
    print("  per frame: wave-cycles %.0f  wait %.0f (%.0f%%)  issue-stall %.0f  issuing %.0f | VALU %.0f SALU %.0f LDS %.1f VMEM rd %.1f wr %.1f" % (
        wc / frames, m.get("SQ_WAIT_ANY", 0) * 4 / frames, 100 * m.get("SQ_WAIT_ANY", 0) * 4 / max(wc, 1),
        m.get("SQ_WAIT_INST_ANY", 0) * 4 / frames, m.get("SQ_ACTIVE_INST_ANY", 0) * 4 / frames,
        m.get("SQ_INSTS_VALU", 0) / frames, m.get("SQ_INSTS_SALU", 0) / frames, m.get("SQ_INSTS_LDS", 0) / frames,
        m.get("SQ_INSTS_VMEM_RD", 0) / frames, m.get("SQ_INSTS_VMEM_WR", 0) / frames))
    if gui:
        simd_cycles = gui * 1024
        print("  kernel %.0f k cycles; of SIMD-cycles: VALU active %.1f%%  LDS-instr active %.1f%%  LDS array busy (IDX_ACTIVE) %.1f%% of CU-cycles, bank conflict %.1f%%" % (
            gui / 1e3, 100 * m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles, 100 * m.get("SQ_ACTIVE_INST_LDS", 0) * 4 / simd_cycles,
            100 * m.get("SQ_LDS_IDX_ACTIVE", 0) / (gui * 256), 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / (gui * 256)))
print(open(os.path.join(out, "p1.log")).read())
PY
find $OUT -name "*.csv" -size +1M -delete
