#!/bin/bash
# Effective shader clock of GEMM variants: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / kernel duration, per dispatch.
# Runs scripts/gemm_freq_run.py (production library, then every scripts/ubench/bin/libgemm_v*.so, 6 launches of 8192^3 each) under
# rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace (counter pass only: never combined with other trace domains).
mkdir -p gpurun_out
export TMPDIR=/tmp
d=/tmp/pmc_freq
rm -rf $d
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $d -o p -- python scripts/gemm_freq_run.py > gpurun_out/freq_run.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/freq_run.log
python - $d <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if 'gemm_kernel' in r['Kernel_Name']:
        vals[int(r['Dispatch_Id'])][r['Counter_Name']] = vals[int(r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
print('dispatch  cycles(GUI_ACTIVE/8)   us      GHz   MFMA_BUSY/(cycles*1024 SIMDs)   raw')
for did in sorted(vals):
    v = vals[did]
    ns, _ = dur.get(str(did), (0, ''))
    cyc = v.get('GRBM_GUI_ACTIVE', 0.0) / 8
    mf = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    print(f'{did:6d} {cyc:12.0f} {ns / 1e3:9.1f} {cyc / max(ns, 1):7.3f}   {mf / max(cyc, 1) / 1024:6.3f}    {dict(v)}')
PY
