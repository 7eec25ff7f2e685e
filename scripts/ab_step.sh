#!/bin/bash
# A/B of two builds of the library inside the training step, on ONE box (boxes differ by up to 10 %): alternates the default build and $1
# usage: scripts/ab_step.sh path/to/other/libaudiolm_hip.so [rounds]
other=$1; rounds=${2:-2}
run() { python bench.py --steps 20 --warmup 5 --residual bf16 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']['all_gemm_launches']['by_kind_ms']
print('$1', d['ms_per_step'], 'ms/step  gemm by kind', r, ' host issue', d['host']['eager_issue_ms'])"; }
for i in $(seq $rounds); do
  run default
  ALM_LIB_PATH=$other run "$other"
done
