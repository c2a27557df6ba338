"""ctypes binding of the C-ABI library (include/vae_step.h).

The product path has no CPU fallback: if the HIP library is missing this module
raises at import of the first symbol, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VAE_STEP_LIB: a diagnostic build of the same library, e.g. `make STAMPS=1 OUT=... OBJD=...` - see tools/diag/gpu_stamps_deep.py)
LIB_PATH = os.environ.get("VAE_STEP_LIB") or os.path.join(_HERE, "lib", "libvae_step_gfx950.so")

NUM_PARAMS = 40
NUM_BN = 8
DTYPE_F32 = 0
DTYPE_BF16 = 1
DTYPE_F16 = 2
COMM_ID_BYTES = 128

PARAM_NAMES = (
    [f"encoder.{i}.{j}" for i in range(4) for j in ("0.weight", "0.bias", "1.weight", "1.bias")]
    + ["fc_mu.weight", "fc_mu.bias", "fc_var.weight", "fc_var.bias", "decoder_input.weight", "decoder_input.bias"]
    + [f"decoder.{i}.{j}" for i in range(3) for j in ("0.weight", "0.bias", "1.weight", "1.bias")]
    + ["final_layer.0.weight", "final_layer.0.bias", "final_layer.1.weight", "final_layer.1.bias",
       "final_layer.3.weight", "final_layer.3.bias"]
)
BN_NAMES = [f"encoder.{i}.1" for i in range(4)] + [f"decoder.{i}.1" for i in range(3)] + ["final_layer.1"]

_lib = None


class VaeLibError(RuntimeError):
    pass


def _sig(fn, res, args):
    fn.restype = res
    fn.argtypes = args


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VaeLibError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C torch_vae_amd/csrc`. There is no CPU fallback for the VAE step.")
    L = C.CDLL(LIB_PATH)
    p, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double
    f64p = C.POINTER(C.c_double)
    i64p = C.POINTER(C.c_int64)
    f32p = C.POINTER(C.c_float)
    _sig(L.vae_last_error, C.c_char_p, [])
    _sig(L.vae_abi_version, i32, [])
    _sig(L.vae_param_layout, i32, [i32, i32, i32, i64p, i64p, i64p])
    _sig(L.vae_bn_layout, i32, [i64p, i64p, i64p])
    _sig(L.vae_create, p, [i32, i32, i32, i32, i32])
    _sig(L.vae_destroy, None, [p])
    _sig(L.vae_workspace_bytes, i64, [p])
    _sig(L.vae_forward, i32, [p, p, i32, p, p, p, p, u64, i32, p, p, p, p, p])
    _sig(L.vae_decode, i32, [p, p, i32, p, p, p, i32, p, p])
    _sig(L.vae_pre_latents, i32, [p, p, p])
    _sig(L.vae_last_eps, i32, [p, p, p])
    _sig(L.vae_loss, i32, [p, f32, p, p])
    _sig(L.vae_loss_deferred, i32, [p, f32, p, p])
    _sig(L.vae_elbo_generic, i32, [p, p, p, p, i64, i32, i32, f32, p, p, p, p, p])
    _sig(L.vae_backward, i32, [p, p, p, p, p, p, p, p, p, p, f32, i32, p])
    _sig(L.vae_backward_part, i32, [p, p, p, p, p, p, p, p, p, p, f32, i32, i32, p])
    _sig(L.vae_comm_stream, i32, [p, p, C.POINTER(C.c_void_p)])
    _sig(L.vae_comm_unique_id, i32, [p])
    _sig(L.vae_comm_init, i32, [p, i32, i32, p])
    _sig(L.vae_comm_world, i32, [p])
    _sig(L.vae_comm_destroy, i32, [p])
    _sig(L.vae_allreduce_grads, i32, [p, p, i32, i64p, i64p, i32, p])
    _sig(L.vae_broadcast_state, i32, [p, p, p, p, i32, p])
    _sig(L.vae_adamw_step, i32, [p, p, p, p, i32, i64p, i64p, f64p, f64p, f64, f64, f64, f32, i32, p])
    _sig(L.vae_train_step, i32, [p, p, i32, p, p, p, p, p, p, p, u64, f32, i32, i64p, i64p, f64p, f64p, f64, f64, f64,
                                 i32, p, p, p, p, p, p])
    _sig(L.vae_train_step_fused, i32, [p, p, i32, p, p, p, p, p, p, p, u64, f32, i32, i64p, i64p, f64p, f64p, f64, f64, f64, f32,
                                       i32, i32, p, p, p, p, p, p])
    _sig(L.vae_synth_pianoroll, i32, [p, i32, i32, u64, p])
    _sig(L.vae_expand_stimuli, i32, [p, i32, p, i64, p])
    _sig(L.vae_profile, i32, [p, i32])
    _sig(L.vae_profile_report, i32, [p, C.c_char_p, i64])
    _sig(L.vae_profile_sequence, i32, [p, C.c_char_p, i64])
    _sig(L.vae_profile_timeline, i32, [p, C.c_char_p, i64])
    _sig(L.vae_debug_stamps, i32, [p, C.c_char_p, i32, p])
    _sig(L.vae_debug_tensor, i32, [p, i32, p, i64, p])
    _sig(L.vae_selftest_tr16, i32, [p])
    _sig(L.vae_set_option, i32, [p, C.c_char_p, i32])
    _lib = L
    return L


EXPORTS = [
    "vae_last_error", "vae_abi_version", "vae_param_layout", "vae_bn_layout", "vae_create", "vae_destroy",
    "vae_workspace_bytes", "vae_forward", "vae_decode", "vae_pre_latents", "vae_last_eps", "vae_loss", "vae_loss_deferred", "vae_elbo_generic",
    "vae_backward", "vae_backward_part", "vae_comm_stream", "vae_comm_unique_id", "vae_comm_init", "vae_comm_world",
    "vae_comm_destroy", "vae_allreduce_grads", "vae_broadcast_state", "vae_adamw_step", "vae_train_step", "vae_train_step_fused", "vae_synth_pianoroll", "vae_expand_stimuli", "vae_profile",
    "vae_profile_report", "vae_profile_sequence", "vae_profile_timeline", "vae_debug_stamps", "vae_debug_tensor",
    "vae_selftest_tr16", "vae_set_option",
]


def check(rc: int, what: str = ""):
    if rc != 0:
        raise VaeLibError(f"{what}: {lib().vae_last_error().decode()}")


def param_layout(img_size: int, latent_dim: int, generalised: bool):
    offs = (C.c_int64 * NUM_PARAMS)()
    sizes = (C.c_int64 * NUM_PARAMS)()
    total = C.c_int64()
    rc = lib().vae_param_layout(img_size, latent_dim, int(generalised), offs, sizes, C.byref(total))
    if rc != 0:
        raise ValueError(lib().vae_last_error().decode())
    return list(offs), list(sizes), total.value


def bn_layout():
    offs = (C.c_int64 * NUM_BN)()
    ch = (C.c_int64 * NUM_BN)()
    total = C.c_int64()
    check(lib().vae_bn_layout(offs, ch, C.byref(total)), "vae_bn_layout")
    return list(offs), list(ch), total.value


def ptr(t):
    """Device pointer of a torch tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()
