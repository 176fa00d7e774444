"""ctypes binding of libld_mi355x.so (C ABI declared in include/ld_mi355x.h).

The product path has NO fallback: if the HIP library is missing, `lib()` raises — it never routes
to PyTorch eager or to the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LD_MI355X_LIB") or os.path.join(_HERE, "libld_mi355x.so")   # env: debug builds only

OK, ERR_ARG, ERR_SHAPE, ERR_HIP, ERR_STATE = 0, 1, 2, 3, 4
F16, F32 = 0, 1


class UNetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("model_channels", C.c_int), ("num_levels", C.c_int),
                ("channel_mult", C.c_int * 8), ("num_res_blocks", C.c_int * 8), ("transformer_depth", C.c_int * 16),
                ("transformer_depth_output", C.c_int * 24), ("transformer_depth_middle", C.c_int),
                ("context_dim", C.c_int), ("num_heads", C.c_int)]


class VAEConfig(C.Structure):
    _fields_ = [("z_channels", C.c_int), ("ch", C.c_int), ("num_levels", C.c_int), ("ch_mult", C.c_int * 8),
                ("num_res_blocks", C.c_int), ("out_ch", C.c_int), ("with_encoder", C.c_int)]


class LDError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = lib().ld_status_string(status).decode() if _lib is not None else str(status)
        super().__init__(f"{where}: {msg} (status {status})")


_lib = None
_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes): every symbol include/ld_mi355x.h declares
SIGNATURES = {
    "ld_version": (C.c_char_p, []),
    "ld_status_string": (C.c_char_p, [_I]),
    "ld_unet_create": (_I, [C.POINTER(UNetConfig), C.POINTER(_P)]),
    "ld_unet_destroy": (None, [_P]),
    "ld_unet_param_count": (_I, [_P]),
    "ld_unet_param_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I), C.POINTER(C.c_int64)]),
    "ld_unet_load_param": (_I, [_P, C.c_char_p, _P, _I, _P]),
    "ld_unet_reserve": (_I, [_P, _I, _I, _I, _I]),
    "ld_unet_workspace_bytes": (_Z, [_P]),
    "ld_unet_weight_bytes": (_Z, [_P]),
    "ld_unet_set_context": (_I, [_P, _P, _I, _I, _I, _P]),
    "ld_unet_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ld_unet_forward_pair": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ld_unet_profile_pair": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I)]),
    "ld_unet_profile": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I)]),
    "ld_unet_profile_kernels": (_I, [_P, C.c_char_p, _Z]),
    "ld_unet_profile_launches": (_I, [_P, C.c_char_p, _Z]),
    "ld_unet_last_launches": (_I, [_P]),
    "ld_unet_last_flops": (C.c_double, [_P]),
    "ld_vae_create": (_I, [C.POINTER(VAEConfig), C.POINTER(_P)]),
    "ld_vae_destroy": (None, [_P]),
    "ld_vae_param_count": (_I, [_P]),
    "ld_vae_param_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I), C.POINTER(C.c_int64)]),
    "ld_vae_load_param": (_I, [_P, C.c_char_p, _P, _I, _P]),
    "ld_vae_reserve": (_I, [_P, _I, _I, _I]),
    "ld_vae_workspace_bytes": (_Z, [_P]),
    "ld_vae_plan_bytes": (_Z, [_P, _I, _I, _I]),
    "ld_vae_decode": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ld_vae_encode": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ld_vae_profile": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ld_vae_profile_launches": (_I, [_P, C.c_char_p, _Z]),
    "ld_vae_last_launches": (_I, [_P]),
    "ld_vae_last_flops": (C.c_double, [_P]),
    "ld_op_linear": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _Z, _P]),
    "ld_op_conv": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "ld_op_groupnorm_conv_ws_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "ld_op_groupnorm_conv": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "ld_op_repack_conv": (_I, [_P, _I, _I, _I, _P, _P]),
    "ld_op_conv_skip_ws_bytes": (_Z, [_I, _I, _I, _I]),
    "ld_op_conv_skip": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "ld_op_groupnorm_ws_bytes": (_Z, [_I, _I]),
    "ld_op_groupnorm": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _F, _I, _P, _P, _P]),
    "ld_op_layernorm": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "ld_op_attention": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "ld_op_conv_gn_partials_floats": (_Z, [_I, _I]),
    "ld_op_conv_gn_partials": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _Z, _P]),
    "ld_op_attention_rowv": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "ld_op_softmax_rows": (_I, [_P, _I, _I, _P]),
    "ld_op_timestep_embed": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "ld_op_cfg_combine": (_I, [_P, _P, _F, _Z, _P]),
    "ld_op_axpby": (_I, [_P, _F, _P, _F, _P, _F, _Z, _P]),
    "ld_op_hook_check": (_I, [_P, _P, _Z, _P, _Z, _P, _I, _P, _I, _P]),
    "ld_op_linear_ln": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _Z, _P]),
    "ld_op_linear_ln_geglu": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _Z, _P]),
    "ld_op_bislerp": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
}


def lib() -> C.CDLL:
    """Load (once) the in-tree HIP library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"or `make -C lightdiffusion_amd/csrc`.  There is no CPU / PyTorch fallback for the hot path.")
        # Runtime resolution order: the library is linked against libamdhip64 without an rpath, so it binds to whichever HIP runtime the
        # process has loaded already, else to the loader's default (/opt/rocm/lib).  PyTorch ships its OWN libamdhip64: when torch is
        # present it must be loaded first — the other way round the process ends up with two HIP runtimes and the first hipMalloc here
        # fails (seen with build() + smoke() in one process).  A pure C-ABI / ctypes host without PyTorch gets the system runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in SIGNATURES.items():
            if not hasattr(l, name):
                if not os.environ.get("LD_MI355X_LIB"):
                    raise AttributeError(f"{LIB_PATH} does not export {name}: header and library out of sync — rebuild it")
                # an older commit's library named for a same-box comparison (tools/build_ref_lib.sh) has fewer entry points: a call of a
                # missing one fails with a clear message instead of an AttributeError deep inside ctypes
                missing.append(name)
                setattr(l, name, _missing_symbol(name))
                continue
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        l._ld_missing = tuple(missing)
        _lib = l
    return _lib


def _missing_symbol(name: str):
    def fail(*_a, **_k):
        raise RuntimeError(f"{LIB_PATH} (LD_MI355X_LIB) does not export {name}: it was built from a commit without this entry point")
    return fail


def check(status: int, where: str) -> None:
    if status != OK:
        raise LDError(status, where)
