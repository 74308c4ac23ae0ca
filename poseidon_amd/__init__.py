"""poseidon_amd — MI355X-native scOT (Poseidon) forward/backward engine.  See DESIGN.md."""
import os as _os

# The engine overlaps two to three HIP streams per step (dependent chain, weight gradients, DP exchange).  The ROCm runtime maps a
# process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and multiplexes the rest: with an RCCL communicator in the
# process (its own streams) the chain and the weight-gradient stream end up sharing a queue and the step loses their overlap —
# measured on MI355X: 27.1 instead of 21.0 ms/step under torch.distributed, back to 21.0 with 8 queues.  The variable is read when
# the HIP runtime initialises, so it has to be in the environment before the first CUDA call of the process (import this package,
# or set it yourself, before touching torch.cuda).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import ScOTConfig, MODEL_MAP, preset  # noqa: F401,E402
