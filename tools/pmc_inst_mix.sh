export TMPDIR=/tmp
R=$PWD
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pe/p$i -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2>&1)
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob("/tmp/pe/p*/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])]+=float(r["Counter_Value"])
    for (_,k,c),v in per.items():
        if "tip::" not in k: continue
        a=acc[k.split("(")[0].replace("void ","")][c]; a[0]+=v; a[1]+=1
for k,v in sorted(acc.items()):
    if any(s in k for s in ("fused_encoder","rnn_rows4","head_ksplit")):
        print(k)
        for c,(s,n) in sorted(v.items()): print(f"    {c:32s} {s/n:.4g}")
PY
