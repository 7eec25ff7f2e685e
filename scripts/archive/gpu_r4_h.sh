#!/bin/bash
# round 4, visit H: where does hc_bwd's time go?  probe builds (scripts/build_variant.sh, ALM_HC_PROBE=1..4), an occupancy-3 build, the two-launch head weight pack
tag=${1:-r4h}
bin=scripts/ubench/bin
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1200 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_${name}.log | cut -c1-500; }
run heads tests/test_gpu_parity.py tests/test_gpu_kernels.py -k "head or logit or pack or hyper_connections"
ALM_LIB_PATH=$bin/libaudiolm_hip_occ3.so run hc_occ3 tests/test_gpu_kernels.py -k "hyper_connections"
for i in 1 2; do
  python scripts/hc_bench.py 2>&1 | tail -1
  ALM_LIB_PATH=$bin/libaudiolm_hip_occ3.so python scripts/hc_bench.py 2>&1 | tail -1
  for p in 1 2 3 4; do ALM_LIB_PATH=$bin/libaudiolm_hip_probe$p.so python scripts/hc_bench.py 2>&1 | tail -1; done
  for o in 1 2 3; do echo -n "occ$o "; ALM_HC_PROBE_OCC=$o ALM_LIB_PATH=$bin/libaudiolm_hip_probe1.so python scripts/hc_bench.py 2>&1 | tail -1; done
  for o in 1; do echo -n "occ$o "; ALM_HC_PROBE_OCC=$o ALM_LIB_PATH=$bin/libaudiolm_hip_probe4.so python scripts/hc_bench.py 2>&1 | tail -1; done
done | tee gpurun_out/${tag}_hc_probe.log
echo "total t=$((SECONDS-t0))"
