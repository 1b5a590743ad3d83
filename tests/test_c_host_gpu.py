"""The drop-in boundary from plain C (examples/c_host.c): a C program that only includes include/tip_hip.h and the HIP runtime —
no Python, no torch — creates a handle, packs weights it generated itself, runs tip_forward and writes y.  The test rebuilds the
same weights and windows in numpy (a counter-based hash), loads them into the Python module and compares: the two hosts drive the
same library, so the outputs must agree bit for bit; and the module's output is checked against the fp64 oracle, so the C host's is."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest
import torch

from tip_amd import synth
from oracle import oracle
from conftest import ROOT
from test_host_cpu import make_model

pytestmark = pytest.mark.gpu
CSRC = os.path.join(ROOT, "transformer-inertial-poser_amd", "csrc")


def _hash32(z):
    z = z.astype(np.uint32)
    z ^= z >> np.uint32(16); z *= np.uint32(0x7FEB352D); z ^= z >> np.uint32(15); z *= np.uint32(0x846CA68B); z ^= z >> np.uint32(16)
    return z


def _unit(stream, n):
    i = np.arange(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = _hash32(i * np.uint32(0x9E3779B1) + _hash32(np.array([stream + 0x1234567], dtype=np.uint32)))
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)


def _weights(m):
    sd = {}
    for i, (name, t) in enumerate(m.state_dict().items()):
        rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.shape[0], 0)
        scale, shift = np.float32(0.05), np.float32(0.0)
        if cols > 0:
            scale, k = np.float32(1.0), 1
            while k * k < cols:
                scale = np.float32(1.0) / np.float32(k + 1)
                k += 1
        if "norm" in name and "weight" in name:
            scale, shift = np.float32(0.1), np.float32(1.0)
        v = shift + scale * _unit(i, t.numel())
        sd[name] = torch.tensor(v.reshape(tuple(t.shape)))
    return sd


@pytest.mark.parametrize("B,T", [(3, 40), (70, 40), (2, 17)])
def test_c_host_matches_python_host(B, T):
    gcc = shutil.which("gcc")
    assert gcc, "the image ships gcc"
    with tempfile.TemporaryDirectory() as td:
        exe, out = os.path.join(td, "c_host"), os.path.join(td, "y.bin")
        cmd = [gcc, "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
               os.path.join(ROOT, "examples", "c_host.c"), "-o", exe, "-L", CSRC, "-ltip_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
               "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe, out, str(B), str(T)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
        assert "forwards=1" in r.stdout
        yc = np.fromfile(out, dtype=np.float32).reshape(B, T, 131)
    cfg = synth.PAPER
    m = make_model(cfg)
    sd = _weights(m)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x_imu = _unit(1000, B * T * 90).reshape(B, T, 90)
    x_s = (np.float32(0.5) * _unit(1001, B * T * 131)).reshape(B, T, 131)
    with torch.no_grad():
        yp = m(torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()).cpu().numpy()
    assert np.isfinite(yc).all()
    assert np.array_equal(yc, yp), np.abs(yc - yp).max()
    w = {k: v.numpy() for k, v in sd.items()}
    yo = oracle.forward(cfg, w, x_imu[:2], x_s[:2], dtype=np.float64)
    assert np.abs(yc[:2] - yo).max() < 2e-5


def test_c_host_reuse_matches_recomputation():
    """tip_forward_reuse from plain C: 60 frames of 3 lock-stepped streams (windows growing 1 .. 40, then sliding) — every call's last
    rows equal what the Python module computes for the same window WITHOUT the ring (two-window encoder for full windows)."""
    gcc = shutil.which("gcc")
    B, F = 3, 60
    with tempfile.TemporaryDirectory() as td:
        exe, out = os.path.join(td, "c_host"), os.path.join(td, "y.bin")
        cmd = [gcc, "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
               os.path.join(ROOT, "examples", "c_host.c"), "-o", exe, "-L", CSRC, "-ltip_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
               "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe, out, str(B), "40", "reuse", str(F)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
        assert f"forwards={F}" in r.stdout
        yc = np.fromfile(out, dtype=np.float32).reshape(F, B, 131)
    m = make_model(synth.PAPER)
    m.load_state_dict(_weights(m))
    m = m.cuda().eval()
    rows_i = np.stack([_unit(2000 + b, F * 90).reshape(F, 90) for b in range(B)])          # [B, F, 90]: one input row per frame
    rows_s = np.stack([np.float32(0.5) * _unit(3000 + b, F * 131).reshape(F, 131) for b in range(B)])
    assert np.isfinite(yc).all()
    with torch.no_grad():
        for f in range(F):
            T = min(f + 1, 40)
            m.set_plan("fused2" if T == 40 else "auto")
            xi = torch.tensor(rows_i[:, f + 1 - T: f + 1]).cuda()
            xs = torch.tensor(rows_s[:, f + 1 - T: f + 1]).cuda()
            yp = m.forward_last(xi, xs).cpu().numpy()
            assert np.array_equal(yc[f], yp), (f, np.abs(yc[f] - yp).max())
