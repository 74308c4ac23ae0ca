"""Smoke run of scOT.trainer.Trainer under torch.distributed (RCCL): `python -m torch.distributed.run --nproc-per-node N
--master-addr 127.0.0.1 tools/dp_trainer_smoke.py [overlap|after]`.  Every rank trains the tiny model on its shard; at the end the
replicas' parameters must be identical and the held-out loss lower than before."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import poseidon_amd  # noqa: E402,F401  (GPU_MAX_HW_QUEUES before the HIP runtime starts)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "overlap"
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import test_model_gpu as T
    from scOT.problems.base import get_dataset
    from scOT.trainer import Trainer, TrainingArguments
    rng = np.random.default_rng(0)
    rd = {"data": np.cumsum(0.2 * rng.standard_normal((12, 21, 5, 32, 32)), axis=1).astype(np.float32)}
    kw = dict(reader=rd, n_max=12, n_val=3, n_test=3, max_num_time_steps=3, time_step_size=2)
    train = get_dataset("fluids.compressible.Riemann", which="train", num_trajectories=6, **kw)
    val = get_dataset("fluids.compressible.Riemann", which="val", num_trajectories=6, **kw)
    train.resolution = val.resolution = 32
    f, meta = T.load_fixture("tiny_trained")
    cfg, model = T.build(meta, "fp16")
    tr = Trainer(model, TrainingArguments(per_device_train_batch_size=8, per_device_eval_batch_size=8, num_train_epochs=2, learning_rate=1e-3,
                                          lr_scheduler_type="cosine", warmup_ratio=0.1, logging_steps=1, max_grad_norm=5.0, dp_exchange=mode),
                 train_dataset=train.to_device(f"cuda:{local}"), eval_dataset=val)
    before = tr.evaluate()["eval_loss"]
    out = tr.train()
    after = tr.evaluate()["eval_loss"]
    flat = model.flat_parameters()
    parts = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    same = all(torch.equal(parts[0], p) for p in parts)
    if rank == 0:
        print(f"dp trainer smoke [{mode}, world {world}]: steps {out.global_step}, eval loss {before:.4f} -> {after:.4f}, replicas identical {same}, "
              f"overflows {int(model._engine.grad_overflow)}")
    assert same and after < before and int(model._engine.grad_overflow) == 0
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
