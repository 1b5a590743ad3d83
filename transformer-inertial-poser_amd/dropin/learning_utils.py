"""Zero-edit drop-in for the reference module of the same name: `from learning_utils import set_seed, loss_q_only_2axis,
loss_constr_multi, loss_jerk` (train_model.py:16), `from learning_utils import set_seed` (offline_testing_simple.py:31,
live_demo_new.py:19, data-gen-and-viz-bullet-new.py:26).  The three losses are tip_amd.learning_utils' (one fused HIP reduction
forward, one gradient kernel backward; device tensors only); set_seed / top_k_logits are host utilities with the reference's
behaviour (learning_utils.py:81-92).  Nothing here imports fairmotion (the reference's import at :9 is unused by these
functions)."""
import _tip_amd_bootstrap

_tip_amd_bootstrap.load()
from tip_amd.learning_utils import (loss_constr_multi, loss_jerk, loss_q_only_2axis, set_seed, top_k_logits,  # noqa: E402,F401
                                    train_loss)
