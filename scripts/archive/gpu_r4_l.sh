#!/bin/bash
# round 4, visit L: hc_bwd LDS-DMA variant -- arithmetic alone at three workgroups per CU (probe 4), and without the second barrier of a token (probe 5)
tag=${1:-r4l}
bin=scripts/ubench/bin
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for i in 1 2 3; do
  python scripts/hc_bench.py 2>&1 | tail -1
  ALM_HC_GL=0 python scripts/hc_bench.py 2>&1 | tail -1
  ALM_LIB_PATH=$bin/libaudiolm_hip_probe4.so python scripts/hc_bench.py 2>&1 | tail -1
  ALM_LIB_PATH=$bin/libaudiolm_hip_probe5.so python scripts/hc_bench.py 2>&1 | tail -1
done | tee gpurun_out/${tag}_hc_probe.log
