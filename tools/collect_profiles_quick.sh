#!/bin/bash
# The round's core measurement artefacts in ~10 GPU-minutes (the full set: tools/collect_profiles.sh):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles_quick.sh r05'
# bench line, rocprofv3 kernel stats (B = 256, 1), PMC passes for the dominant kernel (B = 256), training step, streaming latency.
set -u
ROUND=${1:-r05}
OUT=$PWD/gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
[ -f transformer-inertial-poser_amd/csrc/libtip_hip_measure.so ] || make -C transformer-inertial-poser_amd/csrc -j8 measure > /dev/null
BENCH="python $PWD/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra"
timeout 900 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
for cfg in "256:" "1:--batch 1"; do
    tag=${cfg%%:*}; extra=${cfg#*:}
    d=/tmp/prof_$tag; rm -rf $d
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $BENCH $extra \
        > "$OUT/bench_under_rocprof_B$tag.json" 2> /dev/null)
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_bench_B${tag}_T40.csv"
    t=$(find $d -name '*kernel_trace.csv' | head -1)
    [ -n "$t" ] && python tools/kstats_table.py "$t" > "$OUT/kernel_medians_bench_B${tag}_T40.txt"
    if [ "$tag" = 1 ]; then
        [ -n "$t" ] && python tools/timeline.py "$t" lat_flow_kernel > "$OUT/timeline_B1.txt" 2> /dev/null
    fi
done
PMCS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE")
tag=256
rm -rf /tmp/pmc_$tag; i=0
for grp in "${PMCS[@]}"; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$tag/p$i -- $BENCH --batch $tag > /dev/null 2>&1)
    i=$((i + 1))
done
python - "$tag" "$OUT" <<'PYEOF'
import csv, glob, json, sys, collections
tag, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"/tmp/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, c), v in per.items():
        if "tip::" not in k:
            continue
        a = acc[k.split("(")[0].replace("void ", "")][c]
        a[0] += v
        a[1] += 1
res = {k: {c: {"mean_per_dispatch": s / n, "dispatches": n} for c, (s, n) in sorted(v.items())} for k, v in sorted(acc.items())}
json.dump(res, open(f"{out}/pmc_counters_bench_B{tag}_T40.json", "w"), indent=1)
for k, v in res.items():
    if "fused_encoder" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        t = (2 * v["FETCH_SIZE"]["mean_per_dispatch"] + v["WRITE_SIZE"]["mean_per_dispatch"]) * 1024
        print(f"traffic B{tag} {k}: {t:.0f} bytes/launch")
        json.dump({"kernel": k, "bytes_per_launch": t}, open(f"{out}/traffic_B{tag}.json", "w"))
PYEOF
timeout 600 python tools/train_bench.py > "$OUT/train_bench_n1.json" 2> /dev/null
timeout 600 bash tools/train_profile.sh $ROUND > "$OUT/kernel_avgs_train_B256_T40.txt" 2> /dev/null
timeout 300 python tools/stream_latency.py 1 400 2> /dev/null | grep "^{" > "$OUT/stream_latency_n1.json"
timeout 300 python tools/auto_sweep.py 2> /dev/null > "$OUT/auto_sweep.txt"
# exact streaming reuse: both engines in one run, the front / back end kernels by stream count, the race screen
timeout 400 python tools/reuse_bench.py 2> /dev/null | grep "^{" > "$OUT/reuse_bench.txt"
timeout 400 bash tools/stream_kernels_by_n.sh > "$OUT/stream_kernels_by_n.txt" 2> /dev/null
timeout 300 python tools/reuse_soak.py 1500 2> /dev/null | grep "^{\|^reuse" > "$OUT/reuse_soak.txt"
ls -la "$OUT"

# round 6: the few-stream plan as one launch (stage stamps, one launch vs launch chain, race screen), AUTO's cost model against this box,
# the unedited runner's call (percentiles, growing window), what a kernel boundary / an in-launch hand-off costs on this part
python tools/flow_trace.py 2> /dev/null | grep -v "^model\|^number" > "$OUT/flow_trace_B1.txt"
{ for b in 1 8 16 24 32; do TIP_LAT_FLOW=0 python tools/b1_chain.py $b 2> /dev/null; TIP_LAT_FLOW=64 python tools/b1_chain.py $b 2> /dev/null; done; } > "$OUT/flow_vs_chain.txt"
timeout 600 python tools/flow_soak.py 300 2> /dev/null | grep -v "^model\|^number" > "$OUT/flow_soak.txt"
timeout 600 python tools/auto_calibrate.py --stages 2> /dev/null > "$OUT/auto_calibrate_stages.txt"
timeout 900 python tools/auto_calibrate.py 2> /dev/null > "$OUT/auto_calibrate_sweep.txt"
timeout 300 python tools/zero_edit_loop.py 600 2> /dev/null | grep -v "^model\|^number" > "$OUT/zero_edit_loop.txt"
timeout 300 python tools/zero_edit_breakdown.py 400 2> /dev/null | grep -v "^model\|^number" > "$OUT/zero_edit_breakdown.txt"
for p in l2_probe dataflow_probe; do [ -x tools/probes/$p.out ] && timeout 120 tools/probes/$p.out > "$OUT/$p.txt" 2>&1; done
