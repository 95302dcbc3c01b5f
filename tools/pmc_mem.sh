#!/bin/bash
# Run ON THE GPU BOX: memory-side and SQ counters of the n_fft-1024 kernels, one rocprofv3 --pmc pass per counter group.
# usage: tools/pmc_mem.sh <tag> [perf_all modes]     (environment such as ACIDS_FWD_STORES is inherited)
TAG=${1:-mem}; MODES=${2:-fwd,fused,inv}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_mem_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  echo "pass $i: $GROUP"
  timeout -k 5 150 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/tools/perf_all.py $MODES > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed: $GROUP"
done <<'GROUPS'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY
TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL
TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY TCC_EA0_RDREQ
TCC_REQ TCC_WRITE TCC_HIT TCC_MISS
TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES
TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS
TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ
TCP_TCC_WRITE_REQ_LATENCY TCP_TCP_TA_DATA_STALL_CYCLES
GROUPS
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at_hip::", "")[:78]
        if "stft1024" in k or "istft1024" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
frames = 1024 * 690.0
for k, v in sorted(acc.items()):
    m = {c: sorted(x)[len(x) // 2] for c, x in v.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print("##", k, " kernel %.0f k cycles" % (gui / 1e3))
    for c in sorted(m):
        print("   %-36s %14.0f   per frame %10.2f   per kernel-cycle %8.3f" % (c, m[c], m[c] / frames, m[c] / max(gui, 1)))
PY
find $OUT -name "*.csv" -size +1M -delete
