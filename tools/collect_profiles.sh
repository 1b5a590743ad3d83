#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02'
# Everything lands in gpurun_out/<round>/; copy what should be judged into profiles/<round>/ afterwards.
# PMC passes are run separately from each other and with --kernel-trace only (no sys/hip/hsa trace domains).
set -u
ROUND=${1:-r03}
OUT=$PWD/gpurun_out/$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
# the trace / A-B scripts below load the MEASUREMENT build (TIP_LIB=measure, set by the scripts themselves): the launchers' TIP_*
# environment switches exist only there.  bench.py and the rocprofv3 passes run the default library.
[ -f transformer-inertial-poser_amd/csrc/libtip_hip_measure.so ] || make -C transformer-inertial-poser_amd/csrc -j8 measure > /dev/null
BENCH="python $PWD/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra"

# 1. the default bench line (with cpu_baseline) and the same command under rocprofv3 --kernel-trace --stats
timeout 600 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
for cfg in "256:" "1:--batch 1" "1024:--batch 1024"; do
    tag=${cfg%%:*}; extra=${cfg#*:}
    d=/tmp/prof_$tag; rm -rf $d
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $BENCH $extra \
        > "$OUT/bench_under_rocprof_B$tag.json" 2> /dev/null)
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_bench_B${tag}_T40.csv"
    # rocprofv3's own averages include the first calls of a cold process (round 2: one 27.9-ms outlier in 13 calls moved the
    # B = 1024 encoder's average by 5 %): the same trace with the first fifth of every kernel's calls dropped, min / median / avg
    t=$(find $d -name '*kernel_trace.csv' | head -1)
    [ -n "$t" ] && python tools/kstats_table.py "$t" > "$OUT/kernel_medians_bench_B${tag}_T40.txt"
    if [ "$tag" = 1 ]; then   # kernel timeline of one single-stream forward (latency plan)
        t=$(find $d -name '*kernel_trace.csv' | head -1)
        [ -n "$t" ] && python tools/timeline.py "$t" lat_flow_kernel > "$OUT/timeline_B1.txt" 2> /dev/null
    fi
done

# 2. PMC passes (B=256 and B=1024), one small counter group per pass
PMCS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
      "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
for tag in 256 1024; do
    rm -rf /tmp/pmc_$tag; i=0
    for grp in "${PMCS[@]}"; do
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$tag/p$i -- \
            $BENCH --batch $tag > /dev/null 2>&1)
        i=$((i + 1))
    done
    python - "$tag" "$OUT" <<'EOF'
import csv, glob, json, sys, collections
tag, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"/tmp/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)   # (dispatch, kernel, counter) -> summed over XCD/SE instances
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
# traffic per launch of the dominant kernel: (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, MI355X_MICROARCH.md)
for k, v in res.items():
    if "fused_encoder" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        t = (2 * v["FETCH_SIZE"]["mean_per_dispatch"] + v["WRITE_SIZE"]["mean_per_dispatch"]) * 1024
        print(f"traffic B{tag} {k}: {t:.0f} bytes/launch")
        json.dump({"kernel": k, "bytes_per_launch": t}, open(f"{out}/traffic_B{tag}.json", "w"))
EOF
done

# 3. sweeps
timeout 900 python tools/sweep.py > "$OUT/sweep_n1.jsonl" 2> /dev/null
timeout 600 python tools/stream_bench.py > "$OUT/stream_bench_n1.jsonl" 2> /dev/null
ls -la "$OUT"

# 4. the other rows' benches and the phase / step timelines of the two cooperating kernels
timeout 600 python tools/train_bench.py > "$OUT/train_bench_n1.json" 2> /dev/null
timeout 600 bash tools/train_profile.sh $ROUND > "$OUT/kernel_avgs_train_B256_T40.txt" 2> /dev/null   # + kernel_stats_train_*.csv, timeline_train_step.txt
timeout 300 python tools/loss_bench.py > "$OUT/loss_bench_n1.json" 2> /dev/null
timeout 300 python tools/data_bench.py > "$OUT/data_bench_n1.json" 2> /dev/null
TIP_FUSEDH_TRACE=1 timeout 300 python tools/fh_trace.py 2> /dev/null | grep -v "^model\|^number" > "$OUT/fh_trace_B256.txt"
TIP_RNN_TRACE=1 timeout 300 python tools/rnn_trace.py 2> /dev/null | grep -v "^model\|^number" > "$OUT/rnn_trace_B256.txt"
TIP_RNN_TRACE=1 timeout 300 python tools/rnn_trace.py --cluster 16 2> /dev/null | grep -v "^model\|^number" > "$OUT/rnn_trace_B256_cluster16.txt"
timeout 300 python tools/rnn_variants2.py 2> /dev/null | grep "^B=" > "$OUT/rnn_variants.txt"
{ for a in 0 2; do echo "TIP_RNN_ABLATE=$a"; TIP_RNN_ABLATE=$a timeout 120 python tools/rnn_tsweep.py 256 2> /dev/null | grep "^B=\|^fit"; done; } > "$OUT/rnn_tsweep_B256.txt"
timeout 600 python tools/rnn_soak.py 300 2> /dev/null | grep -v "^model\|^number" > "$OUT/rnn_soak.txt"
timeout 300 python tools/plan_bench.py 256 300 512 1024 2> /dev/null | grep "^B=" > "$OUT/plan_bench.txt"
# round 3: output projection tile stamps / per-workgroup lifetimes, launch-ramp probe, the exploratory split-fp16 plan
TIP_HEAD_TRACE=1 timeout 300 python tools/head_trace.py 256 2> /dev/null | grep -v "^model\|^number" > "$OUT/head_trace_B256.txt"
TIP_HEAD_TRACE=1 timeout 300 python tools/head_trace.py 1024 2> /dev/null | grep -v "^model\|^number" > "$OUT/head_trace_B1024.txt"
timeout 300 python tools/stream_latency.py 1 400 2> /dev/null | grep "^{" > "$OUT/stream_latency_n1.json"
d=/tmp/prof_f16; rm -rf $d
# round 4: AUTO over a batch sweep against the round-3 selection (remainder split, window-split plan), the persistent latency kernel
{ echo "AUTO (round 4)"; timeout 300 python tools/auto_sweep.py 2> /dev/null; echo "round-3 selection (TIP_PLAN_BASE=1)"; TIP_PLAN_BASE=1 timeout 300 python tools/auto_sweep.py 2> /dev/null; } > "$OUT/auto_sweep.txt"
timeout 300 python tools/f1s_bench.py 2> /dev/null | grep "^B=\|fused1s vs" > "$OUT/f1s_bench.txt"
timeout 300 python tools/f1s_parts.py 2> /dev/null | grep "^B=" > "$OUT/f1s_parts.txt"
{ echo "8-wave members"; TIP_RNN_W4=0 timeout 200 python tools/rnn_ab.py 2> /dev/null | grep "^B="; echo "4-wave members (TIP_RNN_W4=1)"; TIP_RNN_W4=1 timeout 200 python tools/rnn_ab.py 2> /dev/null | grep "^B="; } > "$OUT/rnn_w4.txt"
timeout 300 python tools/plan_bench.py 257 272 300 356 1000 2> /dev/null | grep "^B=" > "$OUT/plan_bench_split.txt"
timeout 300 python tools/f64_bench.py 2> /dev/null | grep "^{" > "$OUT/f64_bench_n1.json"
for p in mfma4x4_probe hop_probe permlane_probe launch_probe mfma_f64_probe imul_probe; do [ -x tools/probes/$p.out ] && timeout 120 tools/probes/$p.out > "$OUT/$p.txt" 2>&1; done
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
# exact streaming reuse: both engines in one run, the front / back end kernels by stream count, the race screen
timeout 400 python tools/reuse_bench.py 2> /dev/null | grep "^{" > "$OUT/reuse_bench.txt"
timeout 400 bash tools/stream_kernels_by_n.sh > "$OUT/stream_kernels_by_n.txt" 2> /dev/null
timeout 300 python tools/reuse_soak.py 1500 2> /dev/null | grep "^{\|^reuse" > "$OUT/reuse_soak.txt"
