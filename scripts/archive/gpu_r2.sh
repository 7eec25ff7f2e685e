#!/bin/bash
# Round-2 GPU-box visit.  usage: scripts/gpu_r2.sh <tag> [sections...]   sections: kernels parity fullsize dp rest smoke bench benchbf prof profbf
# Logs -> gpurun_out/<tag>_*.log (merged back by gpurun).
tag=$1; shift
sections="$*"
[[ -z $sections ]] && sections="kernels parity fullsize dp bench benchbf profbf rest smoke"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
export ALM_BENCH_SUPERVISE=0      # rocprofv3 / timing scripts follow ONE process: bench.py measures in place (no re-launching child)
has() { [[ " $sections " == *" $1 "* ]]; }
python - > gpurun_out/${tag}_env.log 2>&1 <<'PY'
import torch, os
print('torch', torch.__version__, 'cuda', torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
p = torch.cuda.get_device_properties(0)
print('CUs', p.multi_processor_count, 'mem GB', p.total_memory / 2**30, 'host cores', os.cpu_count())
PY
cat gpurun_out/${tag}_env.log
t0=$SECONDS
if has kernels; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -n 4 --timeout 300 > gpurun_out/${tag}_kernels.log 2>&1
  echo "kernels rc=$? t=$((SECONDS-t0))"; tail -n 40 gpurun_out/${tag}_kernels.log
fi
if has lab; then
  timeout 900 python -m pytest tests/test_gpu_gemm_lab.py -m gpu -q --tb=short -n 4 --timeout 300 > gpurun_out/${tag}_lab.log 2>&1
  echo "lab rc=$? t=$((SECONDS-t0))"; tail -n 8 gpurun_out/${tag}_lab.log
fi
if has ab; then
  timeout 600 python scripts/ab_gemm.py ${AB_TILES:-2 13 11} > gpurun_out/${tag}_ab_gemm.log 2>&1
  echo "ab rc=$? t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab_gemm.log
fi
if has proto; then
  timeout 600 python -m pytest tests/test_gpu_cache_protocol.py -m gpu -q --tb=short --timeout 300 > gpurun_out/${tag}_proto.log 2>&1
  echo "proto rc=$? t=$((SECONDS-t0))"; tail -n 30 gpurun_out/${tag}_proto.log | cut -c1-400
fi
if has parity; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -s --timeout 600 > gpurun_out/${tag}_parity.log 2>&1
  echo "parity rc=$? t=$((SECONDS-t0))"; grep -E "loss ours|logits|passed|failed|Error|error" gpurun_out/${tag}_parity.log | cut -c1-230 | tail -n 70
fi
if has fullsize; then
  rm -f gpurun_out/r2_fullsize_parity.jsonl
  timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -s --timeout 900 > gpurun_out/${tag}_fullsize.log 2>&1
  echo "fullsize rc=$? t=$((SECONDS-t0))"; grep -E "loss ours|  logits|worst grad|pooled|passed|failed|Error" gpurun_out/${tag}_fullsize.log | cut -c1-230 | tail -n 50
  cp gpurun_out/r2_fullsize_parity.jsonl gpurun_out/${tag}_fullsize_fp32.jsonl 2>/dev/null
fi
if has fullsizebf; then
  rm -f gpurun_out/r2_fullsize_parity.jsonl
  timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -s --timeout 900 -k "bf16" > gpurun_out/${tag}_fullsize_bf16.log 2>&1
  echo "fullsize bf16 rc=$? t=$((SECONDS-t0))"; grep -E "loss ours|  logits|worst grad|pooled|passed|failed|Error" gpurun_out/${tag}_fullsize_bf16.log | cut -c1-230 | tail -n 50
  cp gpurun_out/r2_fullsize_parity.jsonl gpurun_out/${tag}_fullsize_bf16.jsonl 2>/dev/null
fi
if has graphed; then
  timeout 600 python -X faulthandler -m pytest tests/test_gpu_graphed.py -m gpu -q --tb=short --timeout 500 -x > gpurun_out/${tag}_graphed.log 2>&1
  echo "graphed rc=$? t=$((SECONDS-t0))"; tail -n 25 gpurun_out/${tag}_graphed.log | cut -c1-300
fi
if has dp; then
  timeout 600 python -m pytest tests/test_gpu_dp.py -m gpu -q --tb=short --timeout 500 > gpurun_out/${tag}_dp.log 2>&1
  echo "dp rc=$? t=$((SECONDS-t0))"; tail -n 15 gpurun_out/${tag}_dp.log | cut -c1-300
fi
if has bench; then
  timeout 600 python bench.py --steps 20 --warmup 5 --residual fp32 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_bench_fp32.log 2>&1
  echo "bench fp32 rc=$? t=$((SECONDS-t0))"; tail -n 2 gpurun_out/${tag}_bench_fp32.log | cut -c1-1500
fi
if has benchbf; then
  timeout 600 python bench.py --steps 20 --warmup 5 --residual bf16 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_bench_bf16.log 2>&1
  echo "bench bf16 rc=$? t=$((SECONDS-t0))"; tail -n 2 gpurun_out/${tag}_bench_bf16.log | cut -c1-1500
fi
if has sched; then
  for rd in ${SCHED_RD:-bf16 fp32}; do for sc in ${SCHED_SC:-eager graph graph2}; do
    timeout 300 python -X faulthandler bench.py --steps 20 --warmup 5 --residual $rd --schedule $sc --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_sched_${rd}_${sc}.log 2>&1
    echo "sched $rd $sc rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_sched_${rd}_${sc}.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('ms_per_step', 'value', 'loss', 'host')}, d['config'].get('schedule', '')[:40], d['config'].get('schedule_note'))
except Exception as e: print('no json', e)
"
    grep -E "capture failed|Error|error" gpurun_out/${tag}_sched_${rd}_${sc}.log | head -5
  done; done
fi
if has configs; then
  for cf in ${CONFIGS:-coarse1024 fine2049 fine_t2048_q8 e2e_config5}; do
    timeout 900 python bench.py --config $cf --steps 5 --warmup 2 > gpurun_out/${tag}_config_${cf}.log 2>&1
    echo "config $cf rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_config_${cf}.log | cut -c1-2500
  done
fi
if has benchfull; then
  timeout 900 python bench.py > gpurun_out/${tag}_bench_full.log 2>&1
  echo "bench full rc=$? t=$((SECONDS-t0))"; tail -n 2 gpurun_out/${tag}_bench_full.log | cut -c1-3000
fi
prof() {   # $1 = residual dtype, $2 = ALM_ASYNC_WGRAD
  rm -rf /tmp/prof_$1
  ALM_ASYNC_WGRAD=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o r2 -- python bench.py --steps 5 --warmup 2 --residual $1 --schedule eager --no-cpu-baseline --no-optimizer-leg ${BENCH_EXTRA} > gpurun_out/${tag}_prof_$1.log 2>&1
  echo "prof $1 rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_prof_$1.log | cut -c1-300
  db=$(find /tmp/prof_$1 -name "*.db" | head -1)
  if [[ -n $db ]]; then python scripts/prof_summary.py "$db" gpurun_out/${tag}_kernel_stats_$1_async$2.csv "ALM_ASYNC_WGRAD=$2 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --residual $1 --schedule eager --no-cpu-baseline --no-optimizer-leg ${BENCH_EXTRA} (incl. priming + warm-up + 1 instrumented step)"; head -n 28 gpurun_out/${tag}_kernel_stats_$1_async$2.csv | cut -c1-150; fi
}
if has prof; then prof fp32 0; fi
if has profbf; then prof bf16 0; fi
if has pmc; then
  for grp in "FETCH_SIZE:fetch_size" "WRITE_SIZE:write_size" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE:mfma_busy" "SQ_INSTS_VALU_MFMA_MOPS_BF16:mfma"; do
    cnt=${grp%%:*}; nm=${grp##*:}
    ALM_ASYNC_WGRAD=0 bash scripts/pmc.sh "$cnt" ${tag}_$nm > gpurun_out/${tag}_pmc_$nm.out 2>&1
    echo "pmc $nm rc=$? t=$((SECONDS-t0))"; head -n 12 gpurun_out/pmc_${tag}_$nm.csv | cut -c1-220
  done
fi
if has rest; then
  timeout 1500 python -m pytest tests/test_gpu_bias.py tests/test_gpu_codec.py tests/test_gpu_generate.py tests/test_gpu_optimizer.py -m gpu -q --tb=short -n 4 --timeout 600 > gpurun_out/${tag}_rest.log 2>&1
  echo "rest rc=$? t=$((SECONDS-t0))"; tail -n 25 gpurun_out/${tag}_rest.log | cut -c1-300
fi
if has smoke; then
  timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
  echo "smoke rc=$? t=$((SECONDS-t0))"; tail -n 3 gpurun_out/${tag}_smoke.log
fi
echo "total t=$((SECONDS-t0))"
