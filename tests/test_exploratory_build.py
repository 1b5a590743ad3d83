"""The measurement build of the library (csrc: `make measure` -> libtip_hip_measure.so: -DTIP_MEASURE -DTIP_EXPLORATORY) carries what
the default library compiles out: the launchers' TIP_* environment switches and the exploratory split-fp16 plans ("fused16" /
"general16", csrc/tip_s16.hip).  The tests of those plans skip under the default library; here they are re-run in a fresh interpreter
with TIP_LIB=measure, so the exploratory code stays covered without living in the product binary."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

MEASURE_LIB = os.path.join(ROOT, "transformer-inertial-poser_amd", "csrc", "libtip_hip_measure.so")


def _run(args, timeout):
    env = dict(os.environ, TIP_LIB="measure")
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_measurement_build_host_side():
    """CPU: the same ABI (every declared symbol, ABI version) and the packed image's split-fp16 sections."""
    if not os.path.exists(MEASURE_LIB):
        pytest.skip("libtip_hip_measure.so is not built (make -C transformer-inertial-poser_amd/csrc measure)")
    res = _run(["tests/test_host_cpu.py", "-m", "not gpu", "-k", "split_fp16 or pack_options or exports_every_declared"], 600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert " passed" in res.stdout and "skipped" not in res.stdout.splitlines()[-1], res.stdout[-500:]


@pytest.mark.gpu
def test_exploratory_plans_on_the_gpu():
    """MI355X: tests/test_fused16_gpu.py (properties of the split-fp16 plan) and the superseded pair-split plan's parity / fault
    cases against the measurement build."""
    if not os.path.exists(MEASURE_LIB):
        pytest.skip("libtip_hip_measure.so is not built")
    res = _run(["tests/test_fused16_gpu.py", "tests/test_hip_parity.py", "tests/test_handoff_fault_gpu.py", "-m", "gpu",
                "-k", "fused16 or split16 or fused2s or pair_split or keep_mask"], 1500)
    assert res.returncode == 0, res.stdout[-3000:]
    assert " passed" in res.stdout and "skipped" not in res.stdout.splitlines()[-1], res.stdout[-500:]
