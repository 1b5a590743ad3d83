#!/bin/bash
# Measurement: what the recurrence's matrix phase costs without its A-fragment LDS traffic (TIP_RNN_ABLATE=16: 8 instead of 32
# ds_read_b128 per wave and tile, wrong results), next to the no-MFMA (2) and never-wait (1) ablations.  Measurement build only.
for a in 0 16 2 18 1 17; do echo "TIP_RNN_ABLATE=$a"; TIP_RNN_ABLATE=$a python tools/rnn_ab.py 256 1024 2>/dev/null | sed 's/digest.*//'; done
