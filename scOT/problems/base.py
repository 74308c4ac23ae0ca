"""scOT.problems.base — the reference module's import path (reference scOT/problems/base.py:15) for the dataset selector (every reader
of the reference's registry); implementation and the HBM-resident batch assembly: poseidon_amd/data.py."""
from poseidon_amd.data import (BaseDataset, BaseTimeDataset, DeviceTrajectories, PDEDataset, TimePairs, channel_lists,  # noqa: F401
                               get_dataset, resolve_split)
