python -m pytest tests/test_hip_parity.py tests/test_conditioning_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -3
python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'frac', d['whole_forward_frac_of_fp32_mfma_peak'], 'kernel ms', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'frac', d['whole_forward_frac_of_fp32_mfma_peak'], 'kernel ms', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
