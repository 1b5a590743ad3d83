"""tip_amd — MI355X-native hot path of the Transformer Inertial Poser.

One thing lives here: the forward pass of `TF_RNN_Past_State`
(reference: simple_transformer_with_state.py:60-102) as hand-written HIP kernels for gfx950 behind a
C-ABI (`include/tip_hip.h`, built into `csrc/libtip_hip.so`), plus the Python host that mirrors the
reference module surface (`simple_transformer_with_state.TF_RNN_Past_State`).

    import tip_amd
    model = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, ...)   # same signature as the reference
"""
from . import synth  # noqa: F401  (numpy only)

__all__ = ["synth", "TF_RNN_Past_State", "lib", "dist"]


def __getattr__(name):
    # torch / ctypes pieces are imported lazily so that `import tip_amd` works in tooling without torch.
    if name == "TF_RNN_Past_State":
        from .simple_transformer_with_state import TF_RNN_Past_State
        return TF_RNN_Past_State
    if name in ("lib", "dist", "simple_transformer_with_state", "streaming", "data", "learning_utils"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
