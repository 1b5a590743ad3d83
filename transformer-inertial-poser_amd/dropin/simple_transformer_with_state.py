"""Zero-edit drop-in for the reference module of the same name (train_model.py:14, offline_testing_simple.py:80,
live_demo_new.py:16: `from simple_transformer_with_state import TF_RNN_Past_State`).  Put THIS directory — and only this
directory — ahead of the reference's on PYTHONPATH; the class is
tip_amd.simple_transformer_with_state.TF_RNN_Past_State (HIP kernels through libtip_hip.so)."""
import _tip_amd_bootstrap

_tip_amd_bootstrap.load()
from tip_amd.simple_transformer_with_state import TF_RNN_Past_State  # noqa: E402,F401
