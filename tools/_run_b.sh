mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "projection_inside" > gpurun_out/r05b/t1.log 2>&1; echo "t1 rc=$?"; tail -15 gpurun_out/r05b/t1.log
timeout 900 python -m pytest tests/test_handoff_fault_gpu.py -x -q > gpurun_out/r05b/t2.log 2>&1; echo "t2 rc=$?"; tail -5 gpurun_out/r05b/t2.log
for k in 9 4 13 14 8 0 6; do echo "TIP_RNNH_KNOB=$k"; TIP_RNNH_KNOB=$k timeout 300 python tools/rnn_tsweep.py 256 2>/dev/null | tail -3; done > gpurun_out/r05b/knob.txt 2>&1
cat gpurun_out/r05b/knob.txt
python bench.py --no-extra --no-cpu-baseline > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r05b/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['whole_forward_frac_of_fp32_mfma_peak'], d['roofline']['avg_launch_ms'])"
