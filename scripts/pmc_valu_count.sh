#!/bin/bash
# usage: pmc_valu_count.sh <spp> <lib> [<lib> ...]  -> VALU instruction counts of the trace kernel per build (one rocprofv3 --pmc pass each)
spp=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
for lib in "$@"; do
  tag=$(basename $lib .so); OUT=$REPO/gpurun_out/pmcv_$tag; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $OUT/p -o pmc -- python3 $REPO/scripts/pmc_prog.py $REPO/$lib $spp ${PMC_WORKLOAD:-C2} > $OUT/run.log 2> $OUT/run.err ) || { echo "$tag: pass failed"; tail -3 $OUT/run.err; exit 1; }
  python3 - <<PY
import glob, sqlite3
for db in glob.glob("$OUT/p/**/*results.db", recursive=True) + glob.glob("$OUT/p/*results.db"):
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name like '%trace%' group by dispatch_id, counter_name order by dispatch_id"))
    last = max(r[0] for r in rows)
    print("$tag", " ".join(f"{n}={v:.5g}" for d, n, v in rows if d == last))
    break
PY
  tail -1 $OUT/run.log
done
