"""Sensitivity of the step time to one kernel family: the named op wrappers are replaced by no-ops (their outputs stay uninitialised —
TIMING ONLY, never a product path) and the default bench runs.   python tools/whatif.py wgrad_mlp,wgrad_group [bench args]
Groups: side = every weight-gradient / parameter-gradient launch of the side stream."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseidon_amd import ops  # noqa: E402

GROUPS = {"side": "wgrad_mlp,wgrad_group,linear_wgrad,partial_colsum_batch,partial_colsum,cpb_bwd_batched,dwconv7_wgrad,conv5_wgrad,colsum,cln_bwd_finish"}


def main():
    names = sys.argv[1]
    for k, v in GROUPS.items():
        names = names.replace(k, v) if names == k else names
    for n in [x for x in names.split(",") if x and x != "none"]:
        if n.startswith("2x"):       # run an (idempotent) op twice: its marginal cost on the critical path
            f = getattr(ops, n[2:])
            setattr(ops, n[2:], lambda *a, _f=f, **k: (_f(*a, **k), _f(*a, **k))[1])
            continue
        if n == "wgrad_group_deep":     # the grouped weight gradients of the deep stages only (<= 4096 token rows): what a faster TN kernel there could buy at most
            f = ops.wgrad_group
            ops.wgrad_group = lambda cm, problems, _f=f: True if problems[0][0].numel() // problems[0][0].shape[-1] <= 4096 else _f(cm, problems)
            continue
        if not hasattr(ops, n):
            raise SystemExit(f"no op wrapper named {n}")
        ret = True if n in ("wgrad_mlp", "wgrad_group", "block_tail_fwd", "block_tail_bwd") else None
        setattr(ops, n, lambda *a, _r=ret, **k: _r)
    sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--no-parity", "--no-cpu-baseline",
                "--steps", "10", "--warmup", "3"] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
