L=transformer-inertial-poser_amd/csrc/libtip_hip.so
for i in 1 2 3; do for v in old new; do cp tools/probes/_$v.so $L; echo "$v B1024 $(python bench.py --batch 1024 --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")  train $(timeout 300 python tools/train_bench.py --no-composite 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['hip_ms_forward'], d['hip_ms_fwd_bwd'])")"; done; done
cp tools/probes/_new.so $L
python bench.py --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scaled', d['extra']['configs']['scaled_b512_t80'])"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
