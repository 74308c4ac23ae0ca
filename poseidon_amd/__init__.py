"""poseidon_amd — MI355X-native scOT (Poseidon) forward/backward engine.  See DESIGN.md."""
from .config import ScOTConfig, MODEL_MAP, preset  # noqa: F401
