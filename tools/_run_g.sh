#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_latency_gpu.py tests/test_stream_gpu.py tests/test_handoff_fault_gpu.py -m gpu -x -q 2>&1 | tail -5
python bench.py --batch 1 --steps 2000 --warmup 100 --no-cpu-baseline --no-extra
python bench.py --batch 1 --steps 2000 --warmup 100 --no-cpu-baseline --no-extra
python tools/zero_edit_loop.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python /root/repo/bench.py --batch 1 --steps 500 --warmup 50 --no-cpu-baseline --no-extra > /dev/null 2>&1
f=$(find /tmp/p1 -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-200
