"""scOT.trainer — the reference module's import path (reference scOT/trainer.py: `TrainingArguments`, `Trainer`).  The reference's
classes subclass the HF Trainer; these are this library's own driver with the same surface for what the reference's `train.py` /
`inference.py` call (poseidon_amd/train.py: fused arena-wide AdamW, linear / cosine schedule, AR rollout, data parallelism over the
gradient arena, HBM-resident datasets), plus the harness pieces they are built from."""
from poseidon_amd.harness import (compute_loss, conditional_norm_parameter_names, create_optimizer,  # noqa: F401
                                  decay_parameter_names, optimizer_param_groups, rollout)
from poseidon_amd.optim import FusedAdamW  # noqa: F401  (arena-wide AdamW + grad-norm clip: 3 launches per step)
from poseidon_amd.train import (EarlyStoppingCallback, EvalPrediction, PredictionOutput, TrainerCallback, TrainerControl,  # noqa: F401
                                TrainerState, TrainOutput, Trainer, TrainingArguments, checkpoints_in, lr_lambda)
from scOT.model import ConditionalLayerNorm, LayerNorm  # noqa: F401
