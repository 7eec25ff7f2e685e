#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only -- never combined with sys/hip traces) over a short bench.py run.
# usage: scripts/pmc.sh "<counters>" <tag>     -> gpurun_out/pmc_<tag>.csv (per-kernel averages)
cnt="$1"; tag="$2"
mkdir -p gpurun_out
export TMPDIR=/tmp
export ALM_BENCH_SUPERVISE=0      # profilers follow ONE process: bench.py measures in place (no re-launching child)
d=/tmp/pmc_$tag
rm -rf $d
timeout 600 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $d -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optimizer-leg > gpurun_out/pmc_$tag.log 2>&1
echo "rc=$?"
f=$(find $d -name "*counter_collection.csv" | head -1)
echo "file: $f"
python - "$f" gpurun_out/pmc_$tag.csv <<'PY'
import csv, sys, collections
src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
with open(src) as fh:
    for row in csv.DictReader(fh):
        k = row['Kernel_Name'][:110]
        a = agg[k][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
names = sorted({c for k in agg for c in agg[k]})
with open(out, 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'dispatches'] + [n + '_avg' for n in names])
    for k, v in sorted(agg.items(), key=lambda kv: -max(x[0] for x in kv[1].values())):
        n = max(x[1] for x in v.values())
        w.writerow([k, n] + [round(v[c][0] / max(v[c][1], 1), 1) if c in v else '' for c in names])
print(open(out).read()[:6000])
PY
