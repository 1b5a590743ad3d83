L=transformer-inertial-poser_amd/csrc/libtip_hip.so
for i in 1 2 3; do for v in old new; do cp tools/probes/_$v.so $L; echo "$v bench $(python bench.py --no-extra --no-cpu-baseline --steps 400 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"; done; done
cp tools/probes/_new.so $L
TIP_FUSEDH_TRACE=1 timeout 300 python tools/fh_trace.py 2> /dev/null | grep "tail:\|whole window\|prologue\|FFN chunk 1\|out-proj\|head 0"
timeout 900 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pc -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2>&1; python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/pc/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"])]+=float(r["Counter_Value"])
    for (_,k,c),v in per.items():
        if "tip::" in k: acc[(k,c)][0]+=v; acc[(k,c)][1]+=1
for (k,c),(s,n) in sorted(acc.items()): print(k[:60], c, s/n)
PY
