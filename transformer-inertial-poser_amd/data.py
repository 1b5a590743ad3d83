"""Train-set preparation on the GPU: the reference's combiner
(/root/reference/preprocess_and_combine_syn_amass.py:store_imu_s_info) and window dataset
(/root/reference/training_data_loader.py:TrainSubDataset) with the arrays resident in HBM.

    combined = combine_motions(files, down_sample_rates)      # files: dicts unpickled from the reference's .pkl files
    ds = TrainSubDataset.from_arrays(40, combined.info, combined.IMU, combined.S, IMU_sum=combined.SUM)   # re-draw every epoch
    ds = TrainSubDataset(40, info_path, imu_combine_path, s_combine_path, with_acc_sum=True)     # or the reference's files
    for idx in batches_of_indices:
        x_imu, x_s, y = ds.batch(idx)                          # device tensors [n,T,90], [n,T,131], [n,T,131]

`TrainSubDataset` keeps the reference's sampling (python `random`, seeded by the caller) and item protocol
(`ds[i] -> (x_imu, x_s, y_s_n)`, `len(ds)`), so `torch.utils.data.DataLoader(ds, ...)` still works; `batch()` is the fast
path that gathers a whole batch in one kernel.  There is no CPU fallback: tensors must live on the GPU.
"""
from __future__ import annotations

import random
import time
from typing import List, NamedTuple, Optional, Sequence

import numpy as np
import torch

from . import lib as _lib

BIAS_NOISE_ACC = 0.1   # constants.py:19


class Combined(NamedTuple):
    IMU: torch.Tensor     # [N,72] float32, root-local
    SUM: torch.Tensor     # [N,18] float32, running 40-frame acceleration sums / 15
    S: torch.Tensor       # [N,131] float32: 18 x 6D | root velocity | 20 SBP channels
    info: np.ndarray      # rows [start frame, end frame, down-sample rate] (int64)


def combine_motions(files: Sequence[dict], down_sample_rates: Sequence[int], augmented_dip: Optional[Sequence[bool]] = None,
                    biases: Optional[np.ndarray] = None, device="cuda") -> Combined:
    """files[i] = {"imu": [L,72], "nimble_qdq": [L,114], "constrs": [L,20]} (numpy fp64, as in the reference's pickles);
    down_sample_rates[i] / augmented_dip[i] per file (the reference gives them per directory, :36-41).  `biases` [kept,18]
    overrides the accelerometer bias draw (default: np.random.uniform(-0.1, 0.1, 18) per kept file, as :85)."""
    L = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("tip_amd.data: the combiner runs on the GPU only")
    frames = [int(L.tip_combine_frames(len(f["imu"]), len(f["nimble_qdq"]))) for f in files]
    total = sum(frames)
    IMU = torch.empty((total, 72), dtype=torch.float32, device=dev)
    SUM = torch.empty((total, 18), dtype=torch.float32, device=dev)
    S = torch.empty((total, 131), dtype=torch.float32, device=dev)
    scratch = torch.empty(max(frames + [1]) * 18, dtype=torch.float64, device=dev)
    info, start, kept = [], 0, 0
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        for i, f in enumerate(files):
            n = frames[i]
            if n == 0:
                continue                                       # "too short" (:68-70)
            imu = torch.as_tensor(np.ascontiguousarray(f["imu"], dtype=np.float64)).to(dev)
            s = torch.as_tensor(np.ascontiguousarray(f["nimble_qdq"], dtype=np.float64)).to(dev)
            c = torch.as_tensor(np.ascontiguousarray(f["constrs"], dtype=np.float64)).to(dev)
            if abs(len(imu) - len(s)) > 1:
                raise AssertionError("qdq / imu length mismatch")   # :66
            b = biases[kept] if biases is not None else np.random.uniform(-BIAS_NOISE_ACC, BIAS_NOISE_ACC, 18)
            bias = torch.as_tensor(np.asarray(b, dtype=np.float64)).to(dev)
            dip = bool(augmented_dip[i]) if augmented_dip is not None else False
            rc = L.tip_combine_sequence(imu.data_ptr(), s.data_ptr(), c.data_ptr(), len(imu), len(s), bias.data_ptr(), int(dip),
                                        IMU[start:].data_ptr(), SUM[start:].data_ptr(), S[start:].data_ptr(),
                                        scratch.data_ptr(), scratch.numel() * 8, stream)
            if rc != n:
                raise _lib.TipStatusError(rc, "tip_combine_sequence")
            info.append([start, start + n, int(down_sample_rates[i])])
            start += n
            kept += 1
        torch.cuda.current_stream(dev).synchronize()           # the per-file uploads above go out of scope
    return Combined(IMU, SUM, S, np.array(info, dtype=np.int64).reshape(-1, 3))


class TrainSubDataset(torch.utils.data.Dataset):
    """training_data_loader.py:11-92 with the combined arrays resident in HBM.  Same constructor as the reference
    (:19-26: `seq_length, info_path, imu_combine_path, s_combine_path, with_acc_sum`; the acc-sum file is found by the same
    "imu" -> "sum_imu" substitution, :35), same random down-sampling of end frames per epoch (:41-52: python `random`, seeded
    by the caller), same item triple (:72-86).  Windows are gathered on demand instead of being copied up front:

      * `batch(indices)` — the fast path: one gather kernel (tip_gather_windows) from the HBM-resident arrays, device tensors;
      * `ds[i]` / `DataLoader(ds, num_workers=1, pin_memory=True)` (train_model.py:143-147, unedited) — the reference's item
        protocol.  A file-backed dataset answers it with HOST tensors sliced from the memory-mapped .npy files (worker
        processes cannot touch the GPU and `pin_memory` only takes host tensors; a slice is a copy, there is no arithmetic);
        a dataset built from device arrays (`from_arrays`) answers it with device tensors through the same gather kernel.

    `device`: where the combined arrays are uploaded at construction ("cuda" when a GPU is present; None = host item protocol
    only, `batch()` then raises).  There is no CPU gather behind `batch()`."""

    def __init__(self, seq_length, info_path, imu_combine_path, s_combine_path, with_acc_sum=True, *, device="auto"):
        start_time = time.time()
        IMU_c = np.load(imu_combine_path, mmap_mode="r")                                        # :30
        S_c = np.load(s_combine_path, mmap_mode="r")                                            # :31
        infos = np.load(info_path)                                                              # :32
        SUM_c = np.load(imu_combine_path.replace("imu", "sum_imu"), mmap_mode="r") if with_acc_sum else None   # :34-37
        if device == "auto":
            device = "cuda" if torch.cuda.is_available() else None
        self._host = (IMU_c, SUM_c, S_c)
        self._setup(seq_length, infos, with_acc_sum)
        self.IMU_c = self.S_c = self.SUM_c = None
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type != "cuda":
            raise RuntimeError("tip_amd.data.TrainSubDataset: device must be a GPU (or None for the host item protocol only)")
        # The combined arrays go to HBM on the FIRST batch() call, not here: the unedited DataLoader loop of
        # train_model.py builds one dataset per epoch and only ever uses the host item protocol (__getitem__ on the memory maps) — an
        # eager upload was a full read of the files, an H2D copy and several GB of HBM per epoch for nothing.
        # the reference prints the shapes of the windows it materialised (:67-70); same lines, nothing materialised
        n, T = self.size
        print("load time", time.time() - start_time)
        print("IMU shape", torch.Size((n, T, IMU_c.shape[1])) if n else torch.Size((0,)))
        print("IMU sum shape", torch.Size((n, T, SUM_c.shape[1])) if (n and SUM_c is not None) else torch.Size((0,)))
        print("S shape", torch.Size((n, T + 1, S_c.shape[1])) if n else torch.Size((0,)))

    @classmethod
    def from_arrays(cls, seq_length: int, info, IMU: torch.Tensor, S: torch.Tensor, IMU_sum: Optional[torch.Tensor] = None,
                    with_acc_sum: bool = True) -> "TrainSubDataset":
        """From combined arrays already in HBM (`combine_motions(...)`): no files, no host copy."""
        if not (IMU.is_cuda and S.is_cuda):
            raise RuntimeError("tip_amd.data.TrainSubDataset.from_arrays: combined arrays must be on the GPU")
        self = cls.__new__(cls)
        torch.utils.data.Dataset.__init__(self)
        if with_acc_sum and IMU_sum is None:
            raise ValueError("with_acc_sum needs IMU_sum")
        self._host = None
        self.device = IMU.device
        self.IMU_c, self.S_c = IMU.contiguous(), S.contiguous()
        self.SUM_c = IMU_sum.contiguous() if with_acc_sum else None
        self._setup(seq_length, info, with_acc_sum)
        self.ends = self.ends.to(self.device)
        return self

    def _setup(self, seq_length, info, with_acc_sum):
        self.seq_length = int(seq_length)
        self.with_acc_sum = bool(with_acc_sum)
        ends: List[int] = []
        for start_t, end_t, rate in np.asarray(info):
            # each info is [start_t, end_t, down sample] (:44-46)
            time_range = range(int(start_t) + self.seq_length, int(end_t) - 1)
            if len(time_range) == 0:
                continue
            k = int(np.maximum(round(len(time_range) / rate), 1))
            ends += random.sample(time_range, k=k)             # note, set random seed outside (as the reference says)
        self._ends_host = np.asarray(ends, dtype=np.int64)
        self.ends = torch.from_numpy(self._ends_host)
        self.size = (len(ends), self.seq_length)

    def __len__(self):
        return self.size[0]

    def batch(self, index) -> "tuple[torch.Tensor, torch.Tensor, torch.Tensor]":
        """(x_imu [n,T,72(+18)], x_s [n,T,131], y [n,T,131]) for a list / tensor of sample indices: one gather kernel."""
        if self.IMU_c is None and self._host is not None and self.device is not None:
            IMU_c, SUM_c, S_c = self._host     # first use: upload the memory-mapped arrays once
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)   # noqa: E731
            self.IMU_c, self.S_c = up(IMU_c), up(S_c)
            self.SUM_c = up(SUM_c) if SUM_c is not None else None
            self.ends = self.ends.to(self.device)
        if self.IMU_c is None:
            raise RuntimeError("tip_amd.data.TrainSubDataset.batch: the combined arrays are not in HBM (constructed with "
                               "device=None or without a GPU); there is no CPU gather")
        dev = self.IMU_c.device
        idx = torch.as_tensor(index, dtype=torch.int64, device=dev).reshape(-1)
        t = self.ends[idx].contiguous()
        n, T = int(t.numel()), self.seq_length
        wi = int(self.IMU_c.shape[1]) + (int(self.SUM_c.shape[1]) if self.SUM_c is not None else 0)
        ws = int(self.S_c.shape[1])
        if int(self.IMU_c.shape[1]) != 72 or ws != 131 or (self.SUM_c is not None and int(self.SUM_c.shape[1]) != 18):
            raise RuntimeError("tip_amd.data.TrainSubDataset: tip_gather_windows serves the reference's widths (IMU 72, sum 18, S 131)")
        x_imu = torch.empty((n, T, wi), dtype=torch.float32, device=dev)
        x_s = torch.empty((n, T, ws), dtype=torch.float32, device=dev)
        y = torch.empty((n, T, ws), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().tip_gather_windows(self.IMU_c.data_ptr(), self.SUM_c.data_ptr() if self.SUM_c is not None else None,
                                                self.S_c.data_ptr(), int(self.IMU_c.shape[0]), t.data_ptr(), n, T,
                                                x_imu.data_ptr(), x_s.data_ptr(), y.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            raise _lib.TipStatusError(rc, "tip_gather_windows")
        return x_imu, x_s, y

    def __getitem__(self, index):
        if self._host is not None:
            # training_data_loader.py:72-86 on the memory-mapped files: host slices (what a DataLoader worker can deliver)
            IMU_c, SUM_c, S_c = self._host
            t, T = int(self._ends_host[int(index)]), self.seq_length
            x_imu = torch.from_numpy(np.array(IMU_c[t - T:t]))
            if SUM_c is not None:
                x_imu = torch.cat((x_imu, torch.from_numpy(np.array(SUM_c[t - T:t]))), dim=1)
            s = torch.from_numpy(np.array(S_c[t - T:t + 1]))
            return x_imu, s[:-1], s[1:]
        x_imu, x_s, y = self.batch([int(index)])
        return x_imu[0], x_s[0], y[0]

    def __getstate__(self):
        # DataLoader workers (spawn start method) pickle the dataset: device arrays stay with the parent
        st = dict(self.__dict__)
        if self._host is not None:
            st.update(IMU_c=None, S_c=None, SUM_c=None, ends=torch.from_numpy(self._ends_host), device=None)
        return st
