"""scOT.utils — the two model-size helpers the reference's drivers print (reference scOT/utils.py:85-97).  `read_cli` (argparse flags
of the reference's command lines: wandb names, config files) belongs to the CLIs, which are out of scope."""


def get_num_parameters(model) -> int:
    """number of trainable parameters (reference utils.py:85-88)"""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def get_num_parameters_no_embed(model) -> int:
    """... without the embedding and the recovery head (reference utils.py:91-97: names containing "embeddings" / "patch_recovery")"""
    return sum(p.numel() for n, p in model.named_parameters() if p.requires_grad and not ("embeddings" in n or "patch_recovery" in n))
