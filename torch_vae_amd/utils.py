"""Checkpoint helper compatible with the reference's ``utils.safe_save_model`` (utils.py:311-351):
state_dicts of the given modules (+ extra entries) are written to ``.tmp.<name>`` and renamed into place.
The model mirror keeps the reference's state_dict keys, so ``encoder`` / ``decoder`` checkpoints written by
either side load into the other (train.py:320-329, 444-460)."""
from __future__ import annotations

import os

import torch


def safe_save_model(modules, checkpoint_path=None, config=None, **kwargs):
    if checkpoint_path is not None:
        pass
    elif config is not None and hasattr(config, "checkpoint_path"):
        checkpoint_path = config.checkpoint_path
    else:
        raise ValueError("No checkpoint path provided")
    d = os.path.dirname(checkpoint_path)
    if d:
        os.makedirs(d, exist_ok=True)
    tmp_a, tmp_b = os.path.split(checkpoint_path)
    tmp_fname = os.path.join(tmp_a, ".tmp." + tmp_b)
    data = {k: v.state_dict() for k, v in modules.items()}
    data.update(kwargs)
    if config is not None:
        data["config"] = config
    torch.save(data, tmp_fname)
    os.rename(tmp_fname, checkpoint_path)
    return checkpoint_path
