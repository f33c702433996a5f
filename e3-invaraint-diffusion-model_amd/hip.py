"""ctypes binding of the C-ABI HIP library (include/e3d_hip.h -> libe3d_hip.so).

The library is the product's only compute path: there is no CPU or eager-PyTorch fallback.
``lib()`` raises if the shared object is missing (build it with ``__graft_entry__.build()`` or
``csrc/build.sh``).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("E3D_HIP_LIB", os.path.join(_HERE, "libe3d_hip.so"))   # override: kernel experiments
ABI_VERSION = 4

_P = c_void_p
_SIGNATURES = {
    "e3d_abi_version": (c_int, []),
    "e3d_last_error": (c_char_p, []),
    "e3d_gemm_bias_act_f32": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P]),
    "e3d_gemm_bias_act_f32_split": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P]),
    "e3d_gemm_bias_act_f32_split_ex": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P, c_float, _P]),
    "e3d_absmax_f32": (c_int, [_P, c_int64, _P, _P]),
    "e3d_relkey_attn_fwd": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                    _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "e3d_relkey_attn_fwd_split": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                          _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "e3d_gemm_skinny_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "e3d_gemm_skinny_f32_split": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P]),
    "e3d_gemm_skinny_f32_split_ex": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P, c_float, _P]),
    "e3d_gemm_skinny_plan_select": (None, [c_int, c_int]),
    "e3d_gemm_skinny_residual_layernorm_f32_split": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int,
                                                     c_int, _P, c_int64, _P]),
    "e3d_gemm_skinny_residual_layernorm_f32_split_ex": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int,
                                                        c_int, _P, c_int64, c_float, _P]),
    "e3d_attn_skip_padded_tiles": (c_int, [c_int]),
    "e3d_gemm_kernel_select": (c_int, [c_int]),
    "e3d_gemm_general_select": (c_int, [c_int]),
    "e3d_attn_rescale_tau": (c_float, [c_float]),
    "e3d_residual_layernorm_fwd": (c_int, [_P, _P, _P, _P, c_float, _P, _P, c_int, c_int, _P]),
    "e3d_adaln_gate_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P, c_int, c_int, _P]),
    "e3d_embed_layernorm_fwd": (c_int, [_P, c_int, _P, _P, _P, _P, c_float, _P, c_int, _P, _P, c_int, c_int, _P]),
    "e3d_nerf_backbone": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    # training (backward) side
    "e3d_gemm_f32_split_general": (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, _P, _P, c_int64, c_int, c_int,
                                           c_int, c_int, c_int, _P]),
    "e3d_transpose_grouped_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int64, c_int64, _P]),
    "e3d_gemm_wgrad_grouped_f32_split": (c_int, [_P, _P, _P, _P, c_uint64, c_int, c_int64, c_int64, c_int, c_int, c_int, c_int, _P]),
    "e3d_residual_layernorm_drop_fwd": (c_int, [_P, _P, _P, _P, c_float, _P, _P, c_int, c_int, c_float, c_uint64, _P]),
    "e3d_layernorm_bwd_drop": (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_float, c_uint64, _P]),
    # row-complete GEMM + bias + residual + LayerNorm (ABI v4)
    "e3d_weight_planes_bytes": (c_int64, [c_int, c_int]),
    "e3d_weight_planes_f32_split": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "e3d_gemm_residual_layernorm_supported": (c_int, [c_int, c_int, c_int, c_int64]),
    "e3d_gemm_residual_layernorm_f32_split": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P, _P, c_float, _P, c_int64,
                                                      c_int, c_int, c_int, c_int, c_float, _P]),
    "e3d_gemm_wgrad_ragged_f32_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_uint64, c_int, c_int, c_int, _P]),
    "e3d_relkey_attn_bwd_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "e3d_relkey_attn_bwd": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int, _P,
                                    _P, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                    _P, _P, c_int, c_int, c_int, c_int, _P]),
    "e3d_dropout_f32": (c_int, [_P, c_float, c_uint64, _P, c_int64, _P]),
    "e3d_relkey_attn_fwd_split_drop": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                               _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float,
                                               c_uint64, _P]),
    "e3d_relkey_attn_fwd_split_ex": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                             _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float,
                                             c_uint64, _P, c_int, _P, _P, _P, _P]),
    "e3d_attn_scratch_bytes": (c_int64, [c_int]),
    "e3d_ddpm_step_wrap_table": (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_int64, _P]),
    "e3d_relkey_attn_bwd_drop": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int, _P,
                                         _P, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                         _P, _P, c_int, c_int, c_int, c_int, c_float, c_uint64, _P]),
    "e3d_relkey_attn_bwd_ex": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int, _P,
                                       _P, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                       _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_uint64, _P]),
    "e3d_attn_dropout_mask": (c_int, [c_int, c_int, c_int, c_int, c_float, c_uint64, _P, _P]),
    "e3d_layernorm_bwd": (c_int, [_P, _P, _P, c_float, _P, _P, _P, c_int, c_int, _P]),
    "e3d_layernorm_bwd_workspace_floats": (c_int64, [c_int, c_int]),
    "e3d_layernorm_bwd_ws": (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_float, c_uint64, _P, c_int64, _P]),
    "e3d_adaln_gate_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_int, c_int, _P]),
    "e3d_act_fwd": (c_int, [_P, c_int, _P, c_int64, _P]),
    "e3d_act_bwd": (c_int, [_P, _P, c_int, _P, c_int64, _P]),
    "e3d_colsum": (c_int, [_P, c_int64, _P, c_int, c_int, _P]),
    "e3d_group_sum": (c_int, [_P, c_int, _P, c_int, c_int, _P]),
    "e3d_small_k_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "e3d_head_linear_bwd_dx": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "e3d_head_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "e3d_ddpm_step_wrap": (c_int, [_P, _P, _P, c_float, c_float, c_float, c_float, c_int, _P, c_int64, _P]),
    "e3d_q_sample_wrap": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int64, _P]),
    "e3d_discrete_posterior_sample": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P]),
    "e3d_discrete_q_sample": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, _P]),
    # optimizer step (ABI v3)
    "e3d_optim_chunk_elems": (c_int, []),
    "e3d_grad_global_norm": (c_int, [_P, _P, _P, _P, c_int, c_float, _P, _P, _P]),
    "e3d_adamw_step_dyn": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, c_float, c_float, c_float, c_float, _P]),
    "e3d_adamw_step_dev": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "e3d_dropout_set_epoch_ptr": (c_int, [_P]),
    "e3d_adamw_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; fail loudly when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} not found: the HIP extension is the only compute path of this package "
                "(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        # torch ships its own libamdhip64; import it FIRST so that this library binds to the same
        # HIP runtime instance (two runtimes in one process do not share the device context).
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the header and the .so disagree
            fn.restype, fn.argtypes = res, args
        got = handle.e3d_abi_version()
        if got != ABI_VERSION:
            raise HipExtensionMissing(f"{LIB_PATH}: ABI version {got}, binding expects {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().e3d_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
