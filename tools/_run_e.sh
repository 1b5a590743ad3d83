mkdir -p gpurun_out/r05e
python -m pytest tests -m gpu -x -q > gpurun_out/r05e/gputests.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r05e/gputests.log
bash tools/_run_d.sh 2>&1 | grep -v "^model\|^number" | head -12
