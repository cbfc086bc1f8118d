"""ctypes binding of the C-ABI in include/dualdiff_hip.h.

The library is the product: there is no CPU / eager fallback.  If it cannot be loaded the
import of any op fails loudly (RuntimeError) instead of silently running something else.
"""
import ctypes
import os

# torch must be imported BEFORE the library is dlopen'ed: PyTorch-ROCm ships its own libamdhip64
# and the process must end up with ONE HIP runtime (ours resolves to the already-loaded soname);
# loading ours first gives two runtimes and every launch on a torch stream fails.
import torch  # noqa: F401
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p)

from . import _build

DD_F16, DD_BF16, DD_F32 = 0, 1, 2
DD_EPI_NONE, DD_EPI_GEGLU, DD_EPI_SILU = 0, 1, 2


class GemmDesc(Structure):
    _fields_ = [
        ("a", c_void_p), ("a2", c_void_p), ("lda", c_int64), ("lda2", c_int64), ("k1", c_int32),
        ("rows", c_int32), ("n", c_int32), ("k", c_int32),
        ("w", c_void_p), ("bias", c_void_p), ("rowvec", c_void_p),
        ("rows_per_inst", c_int32), ("ld_rowvec", c_int32),
        ("res", c_void_p), ("ldres", c_int64),
        ("out", c_void_p), ("ldc", c_int64),
        ("alpha", c_float), ("accumulate", c_int32), ("epilogue", c_int32),
        ("conv", c_int32), ("hin", c_int32), ("win", c_int32), ("cin", c_int32),
        ("hv", c_int32), ("wv", c_int32), ("hout", c_int32), ("wout", c_int32), ("stride", c_int32),
        ("dtype", c_int32), ("tile", c_int32), ("split_k", c_int32),
        ("ws", c_void_p), ("ws_bytes", c_int64),
        ("ln_colsum", c_void_p), ("ln_bias", c_void_p), ("ln_eps", c_float), ("out_f32", c_int32), ("ln_stats_out", c_void_p), ("ln_stats_in", c_void_p),
        ("out_headmajor_d", c_int32), ("hm_scaled_planes", c_int32), ("hm_scale", c_float),
        ("phase", c_int32),
        ("ln_out", c_void_p), ("ld_ln_out", c_int64), ("lno_gamma", c_void_p), ("lno_beta", c_void_p),
    ]


class AttnDesc(Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p),
        ("ldq", c_int64), ("ldk", c_int64), ("ldv", c_int64), ("ldo", c_int64),
        ("q_batch_stride", c_int64), ("k_batch_stride", c_int64),
        ("v_batch_stride", c_int64), ("o_batch_stride", c_int64),
        ("batch", c_int32), ("heads", c_int32), ("head_dim", c_int32), ("lq", c_int32), ("lk", c_int32),
        ("scale", c_float),
        ("kv_batch_map", c_void_p),
        ("accumulate", c_int32), ("dtype", c_int32), ("variant", c_int32),
        ("q_head_stride", c_int64), ("k_head_stride", c_int64), ("v_head_stride", c_int64),
        ("q_prescaled", c_int32),
        ("kv_batch_map2", c_void_p),
        ("lk_dev", c_void_p),
    ]


class BoxTokensDesc(Structure):
    _fields_ = [
        ("points", c_void_p), ("classes", c_void_p), ("masks", c_void_p),
        ("class_tokens", c_void_p), ("null_pos", c_void_p), ("null_class", c_void_p),
        ("pos", c_void_p), ("cat", c_void_p), ("cls_out", c_void_p),
        ("rows", c_int32), ("points_per_box", c_int32), ("num_freqs", c_int32), ("include_input", c_int32),
        ("class_token_dim", c_int32), ("cls_offset", c_int32),
        ("ld_cat", c_int64),
        ("normalize", c_int32), ("points_dtype", c_int32), ("dtype", c_int32), ("n_classes", c_int32),
        ("freqs", c_float * 16), ("xyz_min", c_float * 3), ("xyz_range", c_float * 3),
    ]


class XAttnDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64), ("res", c_void_p), ("ldres", c_int64),
        ("wq", c_void_p), ("wo", c_void_p), ("bo", c_void_p),
        ("k", c_void_p), ("v", c_void_p), ("ldk", c_int64), ("ldv", c_int64),
        ("k_inst_stride", c_int64), ("k_head_stride", c_int64), ("v_inst_stride", c_int64), ("v_head_stride", c_int64),
        ("out", c_void_p), ("ldo", c_int64),
        ("instances", c_int32), ("rows_per_inst", c_int32), ("lk", c_int32),
        ("channels", c_int32), ("heads", c_int32), ("scale", c_float), ("dtype", c_int32),
        ("ln_out", c_void_p), ("ld_ln_out", c_int64), ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
        ("lk_dev", c_void_p),
    ]


class Gemm8Desc(Structure):
    _fields_ = [
        ("a", c_void_p), ("a_scale", c_void_p), ("lda", c_int64),
        ("w", c_void_p), ("w_scale", c_void_p), ("ldw", c_int64),
        ("bias", c_void_p), ("res", c_void_p), ("ldres", c_int64),
        ("out", c_void_p), ("ldc", c_int64),
        ("rows", c_int32), ("n", c_int32), ("k_padded", c_int32), ("dtype", c_int32),
        ("out_headmajor_d", c_int32), ("hm_scaled_planes", c_int32), ("hm_scale", c_float), ("geglu", c_int32),
    ]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "dd_abi_version": (c_int32, []),
    "dd_desc_size": (c_int64, [c_int32]),
    "dd_error_string": (c_char_p, [c_int32]),
    "dd_target_arch": (c_char_p, []),
    "dd_gemm": (c_int32, [POINTER(GemmDesc), c_void_p]),
    "dd_gemm_workspace_bytes": (c_int64, [POINTER(GemmDesc)]),
    "dd_gemm_num_tiles": (c_int32, []),
    "dd_gemm_tile_id": (c_int32, [c_int32]),
    "dd_gemm_kernel_name": (c_char_p, [POINTER(GemmDesc)]),
    "dd_groupnorm_nhwc": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                    c_int32, c_int32, c_int32, c_float, c_int32, c_int32,
                                    c_void_p, c_int64, c_void_p]),
    "dd_groupnorm_workspace_bytes": (c_int64, [c_int32, c_int32]),
    "dd_groupnorm_is_fused": (c_int32, [c_int32, c_int32, c_int32]),
    "dd_layernorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float,
                               c_int32, c_void_p]),
    "dd_attention": (c_int32, [POINTER(AttnDesc), c_void_p]),
    "dd_attention_kernel_name": (c_char_p, [POINTER(AttnDesc)]),
    "dd_add": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "dd_scale": (c_int32, [c_void_p, c_void_p, c_float, c_int64, c_int32, c_void_p]),
    "dd_probe_spin": (c_int32, [c_void_p, c_uint32, c_void_p]),
    "dd_groupnorm_splitk": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_int32,
                                      c_int32, c_void_p]),
    "dd_xattn320": (c_int32, [POINTER(XAttnDesc), c_void_p]),
    "dd_nchw_to_nhwc_views": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                        c_void_p]),
    "dd_fourier_embed_strided": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, POINTER(c_float), c_int32, c_int32,
                                           c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "dd_box_tokens": (c_int32, [POINTER(BoxTokensDesc), c_void_p]),
    "dd_ctx_assemble": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                  c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dd_gemm8": (c_int32, [POINTER(Gemm8Desc), c_void_p]),
    "dd_rowquant_fp8": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_float,
                                  c_int32, c_void_p]),
    "dd_xattn_pack_weight": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "dd_silu": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "dd_nchw_to_nhwc": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dd_nhwc_to_nchw": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dd_timestep_embedding": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float,
                                        c_int32, c_void_p]),
    "dd_cfg_unipc_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_float, c_int64, c_int32, c_void_p]),
    "dd_softmax_rows": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_int64, c_int32, c_void_p]),
    "dd_ors_project": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                 c_float, c_int32, c_int32, c_int32, c_void_p]),
    "dd_fourier_embed": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, POINTER(c_float), c_int32, c_int32,
                                   c_int32, c_int32, c_void_p]),
    "dd_conv3x3_small_cout": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                        c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dd_conv3x3_thin": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                  c_int32, c_int32, c_int32, c_void_p]),
    "dd_cfg_ddim_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                   c_int64, c_int32, c_void_p]),
}

ABI_VERSION = 4
_LIB = None


def lib_path():
    return _build.lib_path()


def load(build_if_missing=True):
    """Load libdualdiff_hip.so (building it with hipcc when absent and a compiler exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError("dualdiff_amd: %s is missing (run __graft_entry__.build())" % path)
        _build.build_native()
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # fail loudly: there is no fallback path
        raise RuntimeError("dualdiff_amd: cannot load HIP library %s: %s" % (path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RuntimeError("dualdiff_amd: %s does not export %s" % (path, name))
        fn.restype = res
        fn.argtypes = args
    if lib.dd_abi_version() != ABI_VERSION:
        raise RuntimeError("dualdiff_amd: %s has ABI version %d, this binding needs %d (rebuild: __graft_entry__.build())"
                           % (path, lib.dd_abi_version(), ABI_VERSION))
    for which, st in enumerate((GemmDesc, AttnDesc, XAttnDesc, Gemm8Desc, BoxTokensDesc)):
        if lib.dd_desc_size(which) != ctypes.sizeof(st):      # a stale library would read past (or ignore) our fields
            raise RuntimeError("dualdiff_amd: %s was built with a different %s (%d bytes, binding has %d)"
                               % (path, st.__name__, lib.dd_desc_size(which), ctypes.sizeof(st)))
    _LIB = lib
    return lib


class Unsupported(RuntimeError):
    """DD_ERR_UNSUPPORTED (-2): the library validated the call and launched NOTHING — the caller may take another form."""


def check(rc, what):
    if rc != 0:
        msg = load().dd_error_string(rc).decode()
        raise (Unsupported if rc == -2 else RuntimeError)("dualdiff_amd.%s failed: %s (%d)" % (what, msg, rc))
