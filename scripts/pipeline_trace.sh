# rocprofv3 kernel trace of scripts/pipeline_trace.py (twelve 1/8-share frames, two in flight); run on the GPU box from the repo root
set -e
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_pipe; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python3 $REPO/scripts/pipeline_trace.py > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
