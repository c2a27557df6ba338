"""torch_vae_amd: MI355X-native (gfx950) VanillaVAE training step.

Drop-in for the hot path of finlaymiller/torch-vae: ``models.VanillaVAE`` and
``train.train_one_epoch`` with the same call surface, backed by hand-written HIP
kernels behind a C ABI (include/vae_step.h).  Importing this package does not load
the HIP library; the first call that needs it does, and fails loudly if it is missing.
"""
import os as _os

# The step uses five HIP streams (caller's + three side streams + communication); HIP's default of four hardware
# queues makes streams share queues and serialise (+10 % step time once RCCL's streams exist).  Read when HIP starts,
# so it only takes effect if this package is imported before the first HIP call; harmless otherwise.
# With the bucketed gradient exchange (VAE_DP_OVERLAP=1: a sixth stream in flight) EIGHT queues are pathological on MI355X - 2.8 ms
# per step against 1.1 (tools/diag/gpu_dp_exchange.py; independent of the communication stream's priority class) - and six are
# not (1.18 ms), so that mode asks for six; train.fused_step refuses the overlapped exchange when eight are configured.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "6" if _os.environ.get("VAE_DP_OVERLAP") == "1" else "8")

from .types_helpers import EncoderOutput, LossOutput, ModelOutput  # noqa: F401,E402

__all__ = ["VanillaVAE", "FusedAdamW", "train_one_epoch", "build_optimizer", "SyntheticPianorollLoader"]


def __getattr__(name):
    if name == "VanillaVAE":
        from .models import VanillaVAE
        return VanillaVAE
    if name == "FusedAdamW":
        from .optim import FusedAdamW
        return FusedAdamW
    if name == "evaluate":
        from .evaluation import evaluate
        return evaluate
    if name == "safe_save_model":
        from .utils import safe_save_model
        return safe_save_model
    if name in ("train_one_epoch", "build_optimizer", "SyntheticPianorollLoader", "allreduce_gradients", "fused_step"):
        from . import train
        return getattr(train, name)
    raise AttributeError(name)
