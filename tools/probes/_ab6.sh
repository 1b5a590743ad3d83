L=transformer-inertial-poser_amd/csrc/libtip_hip.so
for i in 1 2 3; do for v in old new; do cp tools/probes/_$v.so $L; echo "$v B1 $(python bench.py --batch 1 --no-extra --no-cpu-baseline --steps 2000 --warmup 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; done; done
cp tools/probes/_new.so $L
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -- python /root/repo/bench.py --batch 1 --steps 300 --warmup 20 --no-cpu-baseline --no-extra > /dev/null 2>&1; cd /root/repo; python tools/kstats_table.py $(find /tmp/p1 -name '*kernel_trace.csv' | head -1) | head -4
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_streaming_gpu.py -x -q 2>&1 | tail -2
