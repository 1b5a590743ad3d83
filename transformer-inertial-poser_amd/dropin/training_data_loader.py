"""Zero-edit drop-in for the reference module of the same name (train_model.py:15:
`from training_data_loader import TrainSubDataset`): same constructor (training_data_loader.py:19-26), sampling and item
protocol, combined arrays resident in HBM (tip_amd.data.TrainSubDataset; `batch()` gathers a whole batch in one kernel)."""
import _tip_amd_bootstrap

_tip_amd_bootstrap.load()
from tip_amd.data import TrainSubDataset  # noqa: E402,F401
