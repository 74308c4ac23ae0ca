"""Pins for the dataset front-end's index machinery from the REAL reference base classes (run here only; `scOT/problems/base.py`
imports without h5py): a minimal subclass sets the sizes / label description and calls the reference's own `post_init`, then every
`_idx_map(idx)`, the split bookkeeping and the channel lists are recorded for a grid of settings.  No reference source is restated:
the subclass only supplies attributes the reference's concrete datasets set in their constructors.

usage: python tests/golden/make_dataset_pins.py   ->  tests/golden/dataset_pins.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from scOT.problems.base import BaseDataset, BaseTimeDataset  # noqa: E402  (the reference)


class _T(BaseTimeDataset):
    def __init__(self, n_max, n_val, n_test, label, *a, **k):
        super().__init__(*a, **k)
        self.N_max, self.N_val, self.N_test, self.label_description = n_max, n_val, n_test, label
        self.post_init()


class _S(BaseDataset):
    def __init__(self, n_max, n_val, n_test, label, *a, **k):
        super().__init__(*a, **k)
        self.N_max, self.N_val, self.N_test, self.label_description = n_max, n_val, n_test, label
        self.post_init()


def main():
    cases = []
    for which in ("train", "val", "test"):
        for ntraj in (7, -1, -2, -8):
            for kw in (dict(max_num_time_steps=7, time_step_size=2), dict(max_num_time_steps=3, time_step_size=1),
                       dict(max_num_time_steps=4, time_step_size=2, fix_input_to_time_step=0),
                       dict(max_num_time_steps=5, time_step_size=2, allowed_time_transitions=[1, 3]),
                       dict(max_num_time_steps=3, time_step_size=3, fix_input_to_time_step=2)):
                d = _T(100, 12, 24, "[rho],[u,v],[p],[tracer]", which, ntraj, "./data", None, **kw)
                n = len(d)
                idx = sorted(set(list(range(min(n, 40))) + [n // 2, n - 1]))
                cases.append(dict(kind="time", which=which, num_trajectories=ntraj, kw=kw, length=n, start=d.start, multiplier=d.multiplier,
                                  resolved_trajectories=d.num_trajectories, output_dim=d.output_dim,
                                  channel_slice_list=d.channel_slice_list, descriptors=d.printable_channel_description,
                                  idx=idx, maps=[list(map(int, d._idx_map(i))) for i in idx]))
            s = _S(100, 12, 24, "[u,v],[g]", which, ntraj, "./data", None)
            cases.append(dict(kind="steady", which=which, num_trajectories=ntraj, length=len(s), start=s.start,
                              resolved_trajectories=s.num_trajectories, output_dim=s.output_dim,
                              channel_slice_list=s.channel_slice_list, descriptors=s.printable_channel_description))
    with open(os.path.join(HERE, "dataset_pins.json"), "w") as f:
        json.dump(cases, f)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
