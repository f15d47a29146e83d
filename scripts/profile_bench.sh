#!/bin/bash
# rocprofv3 kernel trace + counter passes of bench.py; run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq1.json 2> $OUT/pmc_sq1.err
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 -d $OUT/pmc_sq2 -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq3 -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq3.json 2> $OUT/pmc_sq3.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
cd $REPO
find $OUT -name "*.csv" | head -50
