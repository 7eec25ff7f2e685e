#!/bin/bash
# round 3, visit H: kernel tests for the hybrid TN plan / hc_param_grads, then A/B of the step: hybrid plan, async prepack
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
t0=$SECONDS
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_graphed.py -m gpu -q --tb=short -x -k "tn or hybrid or hyper or hc or parity or graphed or golden" > gpurun_out/r3h_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/r3h_tests.log | cut -c1-300
run() {
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$*', d['ms_per_step'], d['host'], r['all_gemm_launches']['by_kind_ms'], {k['kernel']: k['avg_launch_us'] for k in r['kernels'] if k['kernel'] in ('tn256','tn128')})"
}
run ALM_GEMM_HYBRID=0 ALM_PREPACK_ASYNC=0
run ALM_GEMM_HYBRID=1 ALM_PREPACK_ASYNC=0
run ALM_GEMM_HYBRID=1 ALM_PREPACK_ASYNC=1
run ALM_GEMM_HYBRID=0 ALM_PREPACK_ASYNC=0
run ALM_GEMM_HYBRID=1 ALM_PREPACK_ASYNC=1
run ALM_GEMM_HYBRID=1 ALM_PREPACK_ASYNC=1 ALM_DEFER_GROUPS=1
echo "total t=$((SECONDS-t0))"
