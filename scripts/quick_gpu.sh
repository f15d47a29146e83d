#!/bin/bash
# usage: quick_gpu.sh <tag> <libA> <libB> ...   : parity subset + A/B timing
tag=$1; shift
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1; echo rc=$? >> gpurun_out/pytest_$tag.log
timeout -k 10 200 python scripts/ab_bench.py --spp 250 "$@" > gpurun_out/ab_$tag.log 2>&1
