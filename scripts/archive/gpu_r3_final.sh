#!/bin/bash
# round 3, final visit: the whole GPU suite (serial), smoke, the driver's default bench line, rocprofv3 kernel stats, PMC passes
tag=${1:-r3z}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
rm -f gpurun_out/r3_fullsize_parity.jsonl gpurun_out/r3_opwise_parity.jsonl
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "all gpu tests rc=$? t=$((SECONDS-t0))"; tail -n 12 gpurun_out/${tag}_gpu_tests.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke rc=$? t=$((SECONDS-t0))"; tail -n 2 gpurun_out/${tag}_smoke.log
timeout 900 python bench.py > gpurun_out/${tag}_bench_default.log 2>&1
echo "bench default rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_default.log | cut -c1-3500
bash scripts/gpu_r3.sh $tag profbf pmc 2>&1 | tail -n 70 | cut -c1-200
python scripts/pmc_summary.py ${tag} gpurun_out/${tag}_pmc_summary.json 2>&1 | tail -n 3
echo "total t=$((SECONDS-t0))"
