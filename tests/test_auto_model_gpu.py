"""GPU: AUTO's plan selection is a cost model in absolute microseconds (csrc/tip_abi.hip), calibrated on one MI355X.  On THIS box:
over the 64-batch sweep of tests/test_benchmarked_shapes_gpu.py no batch size may take more than 10 % longer than a LARGER one (a decision boundary in the wrong place), and
AUTO may not be more than 10 % slower than the best explicit plan that serves the batch.  A clock-capped or
differently binned part that shifts a crossover fails here instead of silently picking slower plans; `python tools/auto_calibrate.py
--stages` then prints the model's constants next to the measured stage costs."""
import os
import sys

import pytest
import torch

from conftest import ROOT
from tip_amd import synth
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_auto_decisions_hold_on_this_box():
    import auto_calibrate
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the cost model is gated to a whole 256-CU part (csrc/tip_abi.hip)")
    m = make_model(synth.PAPER)
    load_synth(m, synth.PAPER, 0)
    m = m.cuda().eval()
    rows, bad = auto_calibrate.sweep(m)
    if bad:                                  # one more look at the offenders only (a neighbour's burst during one timing loop)
        again = sorted({int(b.split("=")[1].split(":")[0].split(" ")[0]) for b in bad})
        near = sorted(set(again + [s for s in auto_calibrate.SIZES for a in again if 0 < s - a <= 64]))
        _, bad = auto_calibrate.sweep(m, sizes=near)
    m.check_handoffs()
    assert not bad, "\n".join(bad)
    assert len(rows) == len(auto_calibrate.SIZES) == 64
