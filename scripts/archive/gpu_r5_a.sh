#!/bin/bash
# round 5, visit A: (1) residual-stream layout micro-benchmark; (2) the new tests (owned scatter, determinism, two forwards per capture);
# (3) owned vs atomic embedding scatter in the step; (4) stall-reason PMC pass of the in-step GEMMs (split the K-step's time)
tag=${1:-r5a}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 120 scripts/ubench/bin/stream_layout > gpurun_out/${tag}_stream_layout.log 2>&1
echo "ubench rc=$? t=$((SECONDS-t0))"; cat gpurun_out/${tag}_stream_layout.log
timeout 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_graphed.py tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider \
  -k "embed_scatter or embed_assemble or two_stack_forwards or redraw or bitwise_deterministic or default_ctor_full_size" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 15 gpurun_out/${tag}_tests.log | cut -c1-300
STEPS=30 bash scripts/ab_env.sh 2 "ALM_EMBED_SCATTER=owned" "ALM_EMBED_SCATTER=atomic" > gpurun_out/${tag}_ab_scatter.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab_scatter.log
bash scripts/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" ${tag}_stall 2>&1 | tail -n 30 | cut -c1-400
echo "pmc1 t=$((SECONDS-t0))"
bash scripts/pmc.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" ${tag}_lds 2>&1 | tail -n 30 | cut -c1-400
echo "pmc2 t=$((SECONDS-t0))"
bash scripts/pmc.sh "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" ${tag}_tcc 2>&1 | tail -n 30 | cut -c1-300
echo "total t=$((SECONDS-t0))"
