#!/bin/bash
# PMC counters of the training step's kernels (paper config), one rocprofv3 pass per counter group (run on the GPU box)
export TMPDIR=/tmp
R=$PWD
PMCS=("SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum")
rm -rf /tmp/tpmc; i=0
for grp in "${PMCS[@]}"; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/tpmc/p$i -- python $R/tools/train_bench.py --no-composite --steps 3 --warmup 1 > /dev/null 2>&1)
    i=$((i + 1))
done
python - "${1:-dwgemm}" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/tpmc/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, c), v in per.items():
        if pat in k:
            a = acc[k.split("(")[0].replace("void ", "")][c]
            a[0] += v; a[1] += 1
for k, v in sorted(acc.items()):
    print(k)
    for c, (s, n) in sorted(v.items()):
        print(f"   {c:32s} {s/n:16.1f}  ({n} dispatches)")
PY
