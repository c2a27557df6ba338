"""Checkpoints in the reference's format.

What is contract (reference utils.py:311-351, train.py:320-329, 444-460) and therefore kept:
  * the call ``safe_save_model(modules, checkpoint_path=None, config=None, **kwargs)``;
  * the file: ONE ``torch.save`` of a dict ``{name: module.state_dict()}`` + the keyword entries + ``config``;
  * the write is atomic: readers see the old file or the complete new one.
How it is written is this package's own: a uniquely named temporary file in the target directory, flushed to
disk, then ``os.replace``d over the target (the reference writes ``.tmp.<name>`` and ``os.rename``s).

The reference's resume logic has two bugs that define what ITS checkpoints contain: only rank
!= 0 ever saves, and ``fc_mu`` / ``fc_var`` / ``decoder_input`` / ``final_layer`` are neither saved nor restored.
``should_save`` and ``checkpoint_modules`` reproduce that by default and fix it behind flags.
"""
from __future__ import annotations

import os
import tempfile

import torch

REFERENCE_MODULES = ("encoder", "decoder")                                   # train.py:447-448
OTHER_MODULES = ("fc_mu", "fc_var", "decoder_input", "final_layer")          # lost by the reference on resume


def _resolve_path(checkpoint_path, config):
    if checkpoint_path is None:
        checkpoint_path = getattr(config, "checkpoint_path", None)
    if checkpoint_path is None:
        raise ValueError("No checkpoint path provided")
    return os.fspath(checkpoint_path)


def safe_save_model(modules, checkpoint_path=None, config=None, **kwargs):
    """Write ``{k: v.state_dict() for k, v in modules}`` + ``kwargs`` (+ ``config``) atomically; returns the path."""
    target = _resolve_path(checkpoint_path, config)
    payload = {name: module.state_dict() for name, module in modules.items()}
    payload.update(kwargs)
    if config is not None:
        payload["config"] = config
    folder = os.path.dirname(target) or "."
    os.makedirs(folder, exist_ok=True)
    fd, scratch = tempfile.mkstemp(prefix="." + os.path.basename(target) + ".", suffix=".partial", dir=folder)
    try:
        with os.fdopen(fd, "wb") as fh:
            torch.save(payload, fh)
            fh.flush()
            os.fsync(fh.fileno())
        os.replace(scratch, target)          # atomic on POSIX: never a half-written checkpoint at `target`
    except BaseException:
        try:
            os.unlink(scratch)
        except OSError:
            pass
        raise
    return target


def should_save(config, fix_rank_gate: bool = False) -> bool:
    """train.py:444: ``config.model_output_dir and (not config.global_rank == 0)`` - rank 0, the only rank of a
    single-process run, never saves.  ``fix_rank_gate=True`` gives the evident intent: rank 0 saves."""
    if not getattr(config, "model_output_dir", None):
        return False
    rank0 = getattr(config, "global_rank", 0) == 0
    return rank0 if fix_rank_gate else not rank0


def checkpoint_modules(model, optimizer, scheduler, save_all_modules: bool = False) -> dict:
    """The ``modules`` argument train.py:445-452 passes; ``save_all_modules=True`` adds the four modules the
    reference drops (their keys are simply absent from reference-written files)."""
    mods = {name: getattr(model, name) for name in REFERENCE_MODULES}
    if save_all_modules:
        mods.update({name: getattr(model, name) for name in OTHER_MODULES})
    mods["optimizer"] = optimizer
    mods["scheduler"] = scheduler
    return mods


def load_checkpoint(model, optimizer, scheduler, checkpoint) -> dict:
    """train.py:320-329: restore encoder / decoder / optimizer / scheduler (+ any of the other four modules present in
    the file); returns ``{"total_step", "n_samples_seen", "epoch", "best_epoch"}``."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(os.fspath(checkpoint), map_location="cpu", weights_only=False)
    for name in REFERENCE_MODULES:
        getattr(model, name).load_state_dict(checkpoint[name])
    for name in OTHER_MODULES:
        if name in checkpoint:
            getattr(model, name).load_state_dict(checkpoint[name])
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint["optimizer"])
    if scheduler is not None:
        scheduler.load_state_dict(checkpoint["scheduler"])
    return {"total_step": checkpoint.get("total_step", 0), "n_samples_seen": checkpoint.get("n_samples_seen", 0),
            "epoch": checkpoint.get("epoch", 0), "best_epoch": checkpoint.get("best_epoch", 0)}
