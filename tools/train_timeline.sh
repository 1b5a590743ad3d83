#!/bin/bash
# kernel timeline of ONE training step under rocprofv3 with the library / environment given by the caller
# usage (GPU box, repo root): [TIP_LIB=measure TIP_DW_GROUP=0 ...] tools/train_timeline.sh <out.txt>
export TMPDIR=/tmp
R=$PWD
OUT=${1:-gpurun_out/train_timeline.txt}
rm -rf /tmp/tt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -- python $R/tools/train_bench.py --no-composite --steps 10 > /dev/null 2>&1)
t=$(find /tmp/tt -name "*kernel_trace.csv" | head -1)
python - "$t" > $R/$OUT <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'tip::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
st = [i for i, r in enumerate(rows) if 'fused_encoder_h_kernel<false, true>' in r['Kernel_Name']]
a, b = st[len(st) // 2], st[len(st) // 2 + 1]
t0 = int(rows[a]['Start_Timestamp']); prev = None
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = (s - prev) / 1000 if prev is not None else 0.0
    g = f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}/{r['Workgroup_Size_X']}"
    print(f"{s/1000:9.1f} dur {(e-s)/1000:7.1f} gap {gap:6.1f} grid {g:>20s}  {r['Kernel_Name'].split('(')[0][:60]}")
    prev = e
print(f"period {(int(rows[b]['Start_Timestamp']) - t0)/1000:.1f} us")
PY
