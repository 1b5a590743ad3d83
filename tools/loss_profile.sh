#!/bin/bash
# per-kernel durations of the fused training losses under rocprofv3 (run on the GPU box from the repo root)
export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/lp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python $R/tools/loss_bench.py $@ > /dev/null 2>&1; f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1); python3 - $f <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "loss_" in r["Name"]:
        print(r["Name"].split("::")[-1].split("(")[0], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
