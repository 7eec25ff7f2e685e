#!/bin/bash
# round 5, round-end sequence on ONE box: the whole GPU suite (serial), smoke, the driver's default bench line, rocprofv3 kernel stats, the PMC passes
# and their summary WITH PROVENANCE (commit + csrc digest: bench.py's roofline.traffic refuses a summary measured on other kernel sources).
# usage: scripts/gpu_r5_final.sh <tag> <commit>        (the box has no .git: pass `git rev-parse --short HEAD`)
tag=${1:-r5z}; commit=${2:-unknown}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
rm -f gpurun_out/r5_fullsize_parity.jsonl gpurun_out/r5_opwise_parity.jsonl
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider --durations=15 > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "all gpu tests rc=$? t=$((SECONDS-t0))"; tail -n 25 gpurun_out/${tag}_gpu_tests.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke rc=$? t=$((SECONDS-t0))"; tail -n 2 gpurun_out/${tag}_smoke.log
timeout 900 python bench.py > gpurun_out/${tag}_bench_default.log 2>&1
echo "bench default rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_default.log | cut -c1-3500
bash scripts/gpu_r3.sh $tag profbf pmc 2>&1 | tail -n 70 | cut -c1-200
python scripts/pmc_summary.py ${tag} gpurun_out/${tag}_pmc_summary.json ${commit} 2>&1 | tail -n 3
# the bench line again, now WITH the PMC summary of this very code in place (roofline.traffic filled from it)
cp gpurun_out/${tag}_pmc_summary.json profiles/r5_pmc_summary.json
timeout 600 python bench.py --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 > gpurun_out/${tag}_bench_with_traffic.log
python - <<PY
import json
d = json.loads(open('gpurun_out/${tag}_bench_with_traffic.log').read())
r = d['roofline']
print('ms/step', d['ms_per_step'], 'traffic', r.get('traffic'), r.get('traffic_source'))
PY
# the documented fallbacks still pass: register-prefetch hc_bwd (ALM_HC_GL=0)
ALM_HC_GL=0 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_opwise.py -m gpu -q --tb=short -p no:cacheprovider -k "hyper_connections or coarse-4-bf16-None" > gpurun_out/${tag}_gl0_tests.log 2>&1
echo "ALM_HC_GL=0 tests rc=$? t=$((SECONDS-t0))"; tail -n 2 gpurun_out/${tag}_gl0_tests.log | cut -c1-200
echo "total t=$((SECONDS-t0))"
# the other BASELINE configurations (each line carries its own roofline, cpu_baseline and oracle loss-parity check)
if [[ -n "$CONFIGS" ]]; then
  rm -f gpurun_out/${tag}_bench_configs.jsonl
  for cf in $CONFIGS; do
    timeout 900 python bench.py --config $cf --steps 5 --warmup 2 2>/dev/null | tail -n 1 >> gpurun_out/${tag}_bench_configs.jsonl
    echo "config $cf rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_configs.jsonl | cut -c1-400
  done
fi
echo "total t=$((SECONDS-t0))"
