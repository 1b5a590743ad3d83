L=transformer-inertial-poser_amd/csrc/libtip_hip.so
for i in 1 2 3; do for v in old new; do cp tools/probes/_$v.so $L; echo "$v $(timeout 300 python tools/train_bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['hip_ms_forward'], d['hip_ms_fwd_bwd'], d['hip_fwd_bwd_frac_fp32_mfma_peak'])")"; done; done
cp tools/probes/_new.so $L
TIP_BWD_TRACE=1 timeout 200 python tools/bwd_trace.py 2>/dev/null | tail -18
timeout 900 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -2
