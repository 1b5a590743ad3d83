#!/bin/bash
# per-kernel averages of the training step under rocprofv3 (run on the GPU box from the repo root): the fused forward / backward kernels
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/tk
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -- python $R/tools/train_bench.py --no-composite --steps 10 > /dev/null 2>&1)
f=$(find /tmp/tk -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tip::' in r['Name'] and float(r['TotalDurationNs']) > 0.4e6:
        print(f"{r['Name'].split('(')[0].replace('void ', '')[:56]:56s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
