L=transformer-inertial-poser_amd/csrc/libtip_hip.so
for i in 1 2 3; do for v in old new; do cp tools/probes/_$v.so $L; echo "$v B1024 $(python bench.py --batch 1024 --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")  B8192 $(python bench.py --batch 8192 --no-extra --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; done; done
cp tools/probes/_new.so $L
timeout 900 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
