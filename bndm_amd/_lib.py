"""ctypes binding of libbndm_hip.so (C ABI declared in include/bndm_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C bndm_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbndm_hip.so")

MAX_LEVELS = 8


class UNetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("resolution", C.c_int),
        ("num_levels", C.c_int), ("block_out_channels", C.c_int * MAX_LEVELS),
        ("down_attn", C.c_int * MAX_LEVELS), ("up_attn", C.c_int * MAX_LEVELS),
        ("layers_per_block", C.c_int), ("dtype", C.c_int), ("max_batch", C.c_int),
    ]


class VaeConfig(C.Structure):
    _fields_ = [
        ("latent_channels", C.c_int), ("out_channels", C.c_int), ("latent_resolution", C.c_int),
        ("num_levels", C.c_int), ("block_out_channels", C.c_int * MAX_LEVELS),
        ("layers_per_block", C.c_int), ("dtype", C.c_int), ("max_batch", C.c_int),
    ]


class UNetProfile(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_conv", C.c_float), ("conv_launches", C.c_int),
                ("conv_flops", C.c_double), ("launches", C.c_int), ("ms_dom", C.c_float),
                ("dom_launches", C.c_int), ("dom_flops", C.c_double), ("dom_bytes", C.c_double)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/bndm_hip.h one to one
SIGNATURES = {
    "bndm_abi_version": (_i, []),
    "bndm_last_error": (C.c_char_p, []),
    "bndm_device_info": (_i, [C.c_char_p, _sz]),
    "bndm_bluenoise_workspace_bytes": (_sz, [_i, _i, _i]),
    "bndm_bluenoise": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "bndm_iadb_step": (_i, [_vp, _vp, _f, _f, _i, _i, _i, _i, _vp]),
    "bndm_vae_decoder_create": (_i, [_vp, _vp]),
    "bndm_vae_decode": (_i, [_vp, _vp, _vp, _i, _vp]),
    "bndm_iadb_train_targets": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _sz, _vp]),
    "bndm_ddim_step": (_i, [_vp, _vp, _f, _f, _f, _f, _f, _sz, _vp]),
    "bndm_export_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "bndm_unet_create": (_i, [C.POINTER(_vp), C.POINTER(UNetConfig)]),
    "bndm_unet_destroy": (None, [_vp]),
    "bndm_unet_num_params": (_i, [_vp]),
    "bndm_unet_param_info": (_i, [_vp, _i, C.c_char_p, _sz, C.POINTER(C.c_int64)]),
    "bndm_unet_load_param": (_i, [_vp, C.c_char_p, _vp, C.c_int64]),
    "bndm_unet_finalize": (_i, [_vp]),
    "bndm_unet_num_ops": (_i, [_vp]),
    "bndm_unet_op_info": (_i, [_vp, _i, C.c_char_p, _sz, C.c_char_p, _sz, C.POINTER(C.c_double)]),
    "bndm_unet_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "bndm_unet_sample_iadb": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bndm_unet_sample_ddim": (_i, [_vp, _vp, _i, _i, _vp, _f, _vp]),
    "bndm_unet_profile": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.POINTER(UNetProfile), _vp]),
}

_lib = None


class BndmError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BndmError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `make -C bndm_amd/csrc` or __graft_entry__.build().")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.bndm_abi_version() != 1:
        raise BndmError("libbndm_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().bndm_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise NotImplementedError(f"{what}: {msg}")
        raise BndmError(f"{what} failed (code {rc}): {msg}")


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t, what):
    if not t.is_cuda:
        raise BndmError(f"{what}: tensor is on {t.device}; this path runs on MI355X only "
                        "(no CPU fallback in the product path)")
