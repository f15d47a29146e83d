#!/bin/bash
# usage: quick_gpu.sh <tag> <libA> <libB> ...   : parity subset + A/B timing (full frame and a 1/8 share)
tag=$1; shift
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1; rc=$?; echo rc=$rc >> gpurun_out/pytest_$tag.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scripts/ab_bench.py --spp 250 "$@" > gpurun_out/ab_$tag.log 2>&1 || exit 1
timeout -k 10 200 python scripts/ab_bench.py --spp 500 --share 8 "$@" >> gpurun_out/ab_$tag.log 2>&1
