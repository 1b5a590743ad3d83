python -m pytest tests/test_train_gpu.py -x -q -k "input_gradients or eval_mode_backward" 2>&1 | grep -E "^E|passed|failed|assert" | head
