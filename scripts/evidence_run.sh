#!/bin/bash
# One evidence set of the current build, run on the GPU box from the repo root: GPU tests, pass fractions of the
# camera rays' first test, bench.py on C2 and C3, rocprofv3 kernel trace and counter passes.  usage: evidence_run.sh <tag>
tag=$1
O=gpurun_out/$tag; mkdir -p $O
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.txt 2>&1 || { echo "gpu tests failed"; tail -20 $O/gpu_tests.txt; exit 1; }
tail -2 $O/gpu_tests.txt
timeout -k 10 200 python scripts/screen_pass_fraction.py > $O/screen_pass_fraction.json 2> $O/screen_pass_fraction.err || { echo "pass fraction failed"; exit 1; }
cp $O/screen_pass_fraction.json profiles/r03_screen_pass_fraction.json
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail $O/bench.err; exit 1; }
timeout -k 10 300 python bench.py --workload C3 --no-cpu-baseline > $O/bench_C3.json 2> $O/bench_C3.err || { echo "bench C3 failed"; exit 1; }
bash scripts/profile_bench.sh > $O/profile_bench.log 2>&1 || { echo "profile failed"; tail $O/profile_bench.log; exit 1; }
python scripts/summarize_rocprof.py gpurun_out/prof_bench $O/final > $O/summarize.log 2>&1 || { echo "summarize failed"; tail $O/summarize.log; exit 1; }
echo evidence ok
