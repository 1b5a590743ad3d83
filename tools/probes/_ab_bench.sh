L=transformer-inertial-poser_amd/csrc/libtip_hip.so
for i in 1 2 3; do for v in old mid new; do cp tools/probes/_$v.so $L; echo "$v $(python bench.py --no-extra --no-cpu-baseline --steps 400 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"; done; done
cp tools/probes/_new.so $L
TIP_FUSEDH_TRACE=1 timeout 300 python tools/fh_trace.py 2> /dev/null | grep "tail:\|whole window\|prologue"
timeout 900 python -m pytest tests/test_hip_parity.py -x -q 2>&1 | tail -2
