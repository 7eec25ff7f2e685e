#!/bin/bash
# the driver's round-end checks at HEAD: whole -m gpu suite (-x), smoke, default bench
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r6z2_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -n 3 gpurun_out/r6z2_gpu_tests.log | cut -c1-200
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -n 1
timeout 600 python bench.py 2>/dev/null | tail -n 1 > gpurun_out/r6z2_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r6z2_bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
