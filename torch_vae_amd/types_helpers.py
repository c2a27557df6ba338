"""Dict shapes returned by the model (mirror of the reference's types_helpers.py:15-37)."""
from typing import TypedDict

from torch import Tensor


class EncoderOutput(TypedDict):
    mu: Tensor
    log_var: Tensor
    pre_latents: Tensor


class ModelOutput(TypedDict):
    output: Tensor
    input: Tensor
    encoded: EncoderOutput
    latents: Tensor


class LossOutput(TypedDict):
    loss: Tensor
    reconstruction_loss: Tensor
    kld_loss: Tensor
