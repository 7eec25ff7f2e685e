#!/bin/bash
# round 4, visit N: hc_bwd LDS-DMA variant with the LDS scalar-record element loop at two workgroups per CU (ALM_HC_GLREC=1 build) vs three-workgroup default
tag=${1:-r4n}
bin=scripts/ubench/bin
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1200 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_${name}.log | cut -c1-400; }
ALM_LIB_PATH=$bin/libaudiolm_hip_glrec.so run hc_glrec tests/test_gpu_kernels.py -k "hyper_connections"
run hc_default tests/test_gpu_kernels.py -k "hyper_connections"
for i in 1 2 3; do
  python scripts/hc_bench.py 2>&1 | tail -1
  ALM_LIB_PATH=$bin/libaudiolm_hip_glrec.so python scripts/hc_bench.py 2>&1 | tail -1
  ALM_HC_GL=0 python scripts/hc_bench.py 2>&1 | tail -1
done | tee gpurun_out/${tag}_hc_ab.log
echo "total t=$((SECONDS-t0))"
