#!/bin/bash
# re-validation of HEAD after the final visit: the whole GPU suite (serial), smoke, the default bench line
tag=${1:-r3zz}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
rm -f gpurun_out/r3_fullsize_parity.jsonl gpurun_out/r3_opwise_parity.jsonl
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "all gpu tests rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_gpu_tests.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_smoke.log
timeout 900 python bench.py > gpurun_out/${tag}_bench_default.log 2>&1
echo "bench default rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_default.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['ms_per_step'], d['value'], d['host'], d['with_optimizer'], d['config']['schedule_probe'], r['frac'], d['parity']['rel'])"
echo "total t=$((SECONDS-t0))"
