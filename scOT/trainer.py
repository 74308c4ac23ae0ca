"""scOT.trainer — harness helpers with the reference module's import path (reference scOT/trainer.py).  The heavy HF
`Trainer` subclass is out of scope; the two behaviours the hot path depends on live in
poseidon_amd.harness and are re-exported here."""
from poseidon_amd.harness import (compute_loss, conditional_norm_parameter_names, create_optimizer,  # noqa: F401
                                  decay_parameter_names, optimizer_param_groups, rollout)
from poseidon_amd.optim import FusedAdamW  # noqa: F401  (arena-wide AdamW + grad-norm clip: 3 launches per step)
from scOT.model import ConditionalLayerNorm, LayerNorm  # noqa: F401
