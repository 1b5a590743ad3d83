#!/bin/bash
# rocprofv3 averages of the streaming front / back end kernels (stream_ingest / stream_consume / reuse_update) by stream count:
#   gpurun -- 'bash tools/stream_kernels_by_n.sh > gpurun_out/stream_kernels_by_n.txt'
export TMPDIR=/tmp
ROOT=$PWD
for n in 1 64 256 1024 4096; do
    d=/tmp/sk_$n; rm -rf $d
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/tools/reuse_bench.py --frames-only $n > /dev/null 2>&1)
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    echo "== n = $n"
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"]
    if any(t in k for t in ("stream_", "reuse_update", "head_", "rnn_", "fused_encoder", "lat_")):
        print(f'{k.split("(")[0].replace("void ", "")[:60]:60s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  min {float(r["MinNs"]) / 1e3:9.1f}')
PY
done
