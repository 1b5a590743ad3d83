#!/usr/bin/env python3
"""Golden output of the REFERENCE train-set combiner and window sampler on synthetic motion
files.  Runs only in the build container: imports /root/reference/preprocess_and_combine_syn_amass.py
(store_imu_s_info) and training_data_loader.py (TrainSubDataset) as they are; fairmotion is stubbed with scipy as in
make_runner_golden.py.  Writes data only: the synthetic inputs, the bias noise the reference drew (np.random seeded),
the combined arrays, the info table, and the windows one seeded epoch samples.

usage: python tests/golden/make_data_golden.py
"""
import os
import pickle
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def synth_motion(L, seed):
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(seed)
    base = Rotation.random(6, random_state=seed)
    w = rng.randn(6, 3) * 0.8
    imu = np.zeros((L, 72))
    acc = rng.randn(6, 3)
    for t in range(L):
        imu[t, :54] = (Rotation.from_rotvec(w * (t / 60.0)) * base).as_matrix().reshape(-1)
        acc = 0.9 * acc + 0.7 * rng.randn(6, 3)
        imu[t, 54:] = acc.reshape(-1)
    s = np.zeros((L, 114))
    s[:, :3] = np.cumsum(rng.randn(L, 3) * 0.01, axis=0)
    s[:, 3:57] = np.cumsum(rng.randn(L, 54) * 0.05, axis=0) + rng.randn(54) * 0.5
    s[:, 57:] = rng.randn(L, 57) * 0.3
    c = np.concatenate([(rng.rand(L, 5, 1) < 0.4) * 1.0, rng.uniform(-0.25, 0.25, (L, 5, 3))], axis=2).reshape(L, 20)
    return imu, s, c


# directory -> [(frames, seed)]; file 1 of each directory loses its last qdq row (the "minor mismatch" of :65-67);
# the 38-frame file is dropped by the reference as too short (:68-70)
LENS = {"syn_A": [(101, 1), (57, 2), (38, 6), (150, 3)], "preprocessed_DIP_IMU_x": [(95, 4), (49, 5)]}
RATES = [7, 3]


def motion_files():
    """[(dir name, file index, imu, s, c)] exactly as written to the pickles."""
    files = []
    for dname, lst in LENS.items():
        for i, (L, seed) in enumerate(lst):
            imu, s, c = synth_motion(L, seed)
            if i == 1:
                s, c = s[:-1], c[:-1]
            files.append((dname, i, imu, s, c))
    return files


def main():
    from make_runner_golden import install_stubs
    install_stubs()
    sys.path.insert(0, "/root/reference")
    import preprocess_and_combine_syn_amass as comb
    from training_data_loader import TrainSubDataset

    comb.name_contains_l = []          # the script's __main__ default (:146)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        dirs, rates = [], RATES
        for dname, i, imu, s, c in motion_files():
            d = os.path.join(tmp, dname)
            if d not in dirs:
                os.makedirs(d)
                dirs.append(d)
            with open(os.path.join(d, f"m{i}.pkl"), "wb") as f:
                pickle.dump({"imu": imu, "nimble_qdq": s, "constrs": c}, f)
        k = len(motion_files())
        np.random.seed(1234)
        state = np.random.get_state()
        biases = []
        for _ in range(k):                            # the reference draws one uniform(-0.1, 0.1, 18) per kept file (:85)
            biases.append(np.random.uniform(-0.1, 0.1, 18))
        out["biases"] = np.array(biases)
        np.random.set_state(state)
        imu_path, s_path, info_path = (os.path.join(tmp, n) for n in ("imu_train_t", "s_train_t", "info_train_t"))
        comb.store_imu_s_info(dirs, rates, imu_path, s_path, info_path, num_sbps=5)
        out["IMU"] = np.load(imu_path + ".npy")
        out["SUM"] = np.load(imu_path.replace("imu", "sum_imu") + ".npy")
        out["S"] = np.load(s_path + ".npy")
        out["info"] = np.load(info_path + ".npy")
        random.seed(99)
        ds = TrainSubDataset(40, info_path + ".npy", imu_path + ".npy", s_path + ".npy", with_acc_sum=True)
        out["n_windows"] = np.array([len(ds)])
        xs = [ds[i] for i in range(len(ds))]
        out["win/x_imu"] = np.stack([x[0].numpy() for x in xs[:3]])          # three windows in full ...
        out["win/x_s"] = np.stack([x[1].numpy() for x in xs[:3]])
        out["win/y"] = np.stack([x[2].numpy() for x in xs[:3]])
        out["win/sums"] = np.array([[np.nansum(x[0].numpy().astype(np.float64)), np.nansum(x[1].numpy().astype(np.float64)),
                                     np.nansum(x[2].numpy().astype(np.float64))] for x in xs])   # ... a checksum of every one
    path = os.path.join(HERE, "tip_data_golden.npz")
    np.savez_compressed(path, **out)
    print("IMU", out["IMU"].shape, "S", out["S"].shape, "info", out["info"].tolist(), "windows", out["win/sums"].shape)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
