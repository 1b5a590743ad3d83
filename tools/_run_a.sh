mkdir -p gpurun_out/r05a
for a in 0 16 32 48; do echo "TIP_RNN_ABLATE=$a"; TIP_RNN_ABLATE=$a python tools/rnn_tsweep.py 256 2>/dev/null; done > gpurun_out/r05a/rnn_hopfill_probe.txt 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05a/gputests.log
python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r05a/gputests.log; cat gpurun_out/r05a/rnn_hopfill_probe.txt
