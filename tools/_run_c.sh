mkdir -p gpurun_out/r05c
for k in 3 7 1027 1029 1031; do echo "== TIP_RNNH_KNOB=$k"; TIP_RNNH_KNOB=$k timeout 300 python tools/rnn_tsweep.py 256 2>/dev/null | tail -2;  TIP_RNNH_TRACE=1 TIP_RNNH_KNOB=$k timeout 300 python tools/rnnh_trace.py 2>/dev/null; done > gpurun_out/r05c/trace.txt 2>&1
cat gpurun_out/r05c/trace.txt
