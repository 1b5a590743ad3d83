#!/bin/bash
# per-launch durations of the weight-gradient GEMMs of the scaled configuration under rocprofv3 (run on the GPU box)
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/tps
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tps -- python $R/tools/train_bench.py --config scaled --batch ${1:-64} --steps 2 --warmup 1 --no-composite > /dev/null 2>&1)
t=$(find /tmp/tps -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'dwgemm' in n or 'tgemm_kernel<1, 1' in n or 'splitk' in n:
        agg[(n, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    print(f"{k[0][:40]:40s} grid {k[1]:>8s}x{k[2]}x{k[3]} calls {len(v):4d} avg {sum(v)/len(v):9.1f} us")
PY
