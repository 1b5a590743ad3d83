#!/bin/bash
# rocprofv3 kernel stats (min / median / average duration) of the tip:: kernels of one command.
# usage: tools/kstats.sh <tag> -- <command...>      -> gpurun_out/kstats_<tag>.txt
set -u
TAG=$1; shift; shift
export TMPDIR=/tmp
ROOT=$PWD
d=/tmp/kst_$TAG; rm -rf $d
(cd /tmp && PYTHONPATH=$ROOT timeout 900 rocprofv3 --kernel-trace --output-format csv -d $d -- "$@" > /dev/null 2>&1)
t=$(find $d -name '*kernel_trace.csv' | head -1)
mkdir -p $ROOT/gpurun_out
python - "$t" <<'PY' | tee $ROOT/gpurun_out/kstats_$TAG.txt
import csv, sys, collections, statistics
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tip::" not in k: continue
    rows[k.split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':70s} {'calls':>6s} {'min_us':>9s} {'median_us':>9s} {'avg_us':>9s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v) // 5:]          # drop the first fifth (warm-up / clock ramp)
    print(f"{k[:70]:70s} {len(v):6d} {min(v):9.2f} {statistics.median(v):9.2f} {sum(v) / len(v):9.2f}")
PY
