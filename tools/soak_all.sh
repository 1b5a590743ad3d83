mkdir -p gpurun_out/r06s
O=gpurun_out/r06s
timeout 400 python tools/fuzz_parity.py 300 5 2>/dev/null | tail -6 > $O/fuzz_parity.txt
timeout 300 python tools/fuzz_train.py 200 5 2>/dev/null | tail -6 > $O/fuzz_train.txt
timeout 300 python tools/fuzz_stream.py 150 5 2>/dev/null | tail -6 > $O/fuzz_stream.txt
timeout 400 python tools/rnn_soak.py 300 2>/dev/null | grep -v "^model\|^number" | tail -12 > $O/rnn_soak.txt
timeout 300 python tools/latency_soak.py 2>/dev/null | tail -8 > $O/latency_soak.txt
timeout 300 python tools/f1s_soak.py 2>/dev/null | tail -8 > $O/f1s_soak.txt
timeout 300 python tools/train_soak.py 2>/dev/null | tail -8 > $O/train_soak.txt
timeout 300 python tools/options_soak.py 1 2000 2>/dev/null | tail -6 > $O/options_soak.txt
timeout 300 python tools/reuse_soak.py 1500 2>/dev/null | tail -6 > $O/reuse_soak.txt
for f in $O/*.txt; do echo "== $f"; cat $f; done
