#!/bin/bash
# usage: pmc_passes.sh <tag> <lib> <spp> [workload]  -> gpurun_out/pmc_<tag>/summary.txt
tag=$1; lib=$2; spp=$3; wl=${4:-C2}
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$tag; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_IFETCH SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_INT64 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $OUT/p$i -o pmc -- python3 $REPO/scripts/pmc_prog.py $REPO/$lib $spp $wl > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/fail.txt
done
cd $REPO
python3 - <<PY > $OUT/summary.txt
import glob, sqlite3
vals = {}
for db in sorted(glob.glob("$OUT/p*/pmc_results.db")):
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name like '%trace%' group by dispatch_id, counter_name order by dispatch_id"))
    last = max(r[0] for r in rows)
    for d, n, v in rows:
        if d == last: vals[n] = v
for k in sorted(vals): print(f"{k},{vals[k]:.6g}")
PY
cat $OUT/p1.log | tail -1 >> $OUT/summary.txt
cat $OUT/summary.txt
