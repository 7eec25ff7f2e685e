#!/bin/bash
# Round-end validation on one GPU box, as the driver does it: the whole -m gpu suite, smoke(), the default bench line, then the committed evidence
# (kernel-trace profile of the headline step, the other BASELINE configs).  usage: scripts/gpu_final.sh <tag>
tag=$1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 900 > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "gpu tests rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_gpu_tests.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke rc=$? t=$((SECONDS-t0))"; tail -n 2 gpurun_out/${tag}_smoke.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/${tag}_bench.log 2>&1
echo "bench rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench.log | cut -c1-3500
bash scripts/gpu_r2.sh ${tag} profbf configs 2>&1 | cut -c1-2600
echo "total t=$((SECONDS-t0))"
