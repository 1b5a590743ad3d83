#!/bin/bash
# kernel breakdown of the training step under rocprofv3 (run on the GPU box from the repo root)
export TMPDIR=/tmp
R=$PWD
ROUND=${1:-r02}
rm -rf /tmp/tp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python $R/tools/train_bench.py --no-composite --steps 10 > /dev/null 2>&1)
f=$(find /tmp/tp -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/$ROUND
cp $f $R/gpurun_out/$ROUND/kernel_stats_train_B256_T40.csv
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tip::' in r['Name']:
        n = r['Name'].split('(')[0].replace('void ', '')
        print(f"{n:56s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
t=$(find /tmp/tp -name "*kernel_trace.csv" | head -1)
python - "$t" > $R/gpurun_out/$ROUND/timeline_train_step.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'tip::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def step_starts():
    # a step begins with the weight-image pack launches in front of the training forward's encoder kernel (fused path), or with
    # prep_in_kernel (layer-by-layer path)
    st = [i for i, r in enumerate(rows) if 'prep_in_kernel' in r['Kernel_Name']]
    if st:
        return st
    for i, r in enumerate(rows):
        if 'fused_encoder_h_kernel<false, true>' in r['Kernel_Name'] or 'fused_encoder_kernel<8' in r['Kernel_Name']:
            j = i
            while j > 0 and 'pack_ops_kernel' in rows[j - 1]['Kernel_Name']:
                j -= 1
            st.append(j)
    return st
idx = step_starts()
a, b = idx[-2], idx[-1]          # one full step (forward + backward) near the end
t0 = int(rows[a]['Start_Timestamp'])
prev = None
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = (s - prev) / 1000 if prev is not None else 0.0
    g = f"{r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}"
    print(f"{s/1000:9.1f} dur {(e-s)/1000:7.1f} gap {gap:6.1f} grid {g:>16s}  {r['Kernel_Name'].split('(')[0].replace('void ','')[:60]}")
    prev = e
PY
tail -130 $R/gpurun_out/$ROUND/timeline_train_step.txt
