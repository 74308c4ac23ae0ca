"""Drop-in `scOT` package: same import path and public names as the reference (`from scOT.model import ScOT, ScOTConfig`),
backed by the MI355X-native engine in `poseidon_amd`."""
