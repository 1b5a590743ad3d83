for i in 1 2 3; do python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'frac', d['whole_forward_frac_of_fp32_mfma_peak'], 'kernel ms', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['launches_timed'])"; done
python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -2
