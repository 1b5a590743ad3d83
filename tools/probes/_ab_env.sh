for i in 1 2 3; do for v in 0 1; do echo "TIP_FFN_BWD_STAGGER=$v $(TIP_FFN_BWD_STAGGER=$v bash tools/train_kernels.sh 2>/dev/null | grep "ffn_bwd")"; done; done
TIP_BWD_TRACE=1 timeout 200 python tools/bwd_trace.py 2>/dev/null | head -9
timeout 900 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -2
