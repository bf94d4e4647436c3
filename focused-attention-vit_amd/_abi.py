"""ctypes binding of libfavit.so (the C ABI declared in include/favit.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C focused-attention-vit_amd/csrc``.  There is NO fallback: if the library is
missing every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfavit.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "favit.h")

F32, BF16, FP8 = 0, 1, 2
E4M3, E5M2 = 0, 1
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_GELU_SAVEGRAD, ACT_MULAUX = 0, 1, 2, 3, 4
ERR_INVALID, ERR_UNSUPPORTED, ERR_ALIGN, ERR_LAUNCH = -1, -2, -3, -4
POOL = {"mean": 0, "max": 1, "attention": 2}

vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float


class GemmDesc(C.Structure):
    """Mirror of favit_gemm_t (include/favit.h)."""
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("aux_in", vp), ("aux_out", vp), ("residual", vp),
        ("a_rowsum", vp),
        ("M", i64), ("N", i64), ("K", i64),
        ("lda", i64), ("ldb", i64), ("ldc", i64), ("ld_aux_in", i64), ("ld_aux_out", i64), ("ld_res", i64),
        ("sAo", i64), ("sAi", i64), ("sBo", i64), ("sBi", i64), ("sCo", i64), ("sCi", i64),
        ("batch", i32), ("batch_inner", i32), ("a_kmajor", i32), ("b_kmajor", i32),
        ("in_dtype", i32), ("out_dtype", i32), ("act", i32), ("accumulate", i32), ("split_k", i32),
        ("alpha", f32), ("dropout_p", f32), ("fp8_fmt", i32), ("dropout_seed", u64),
        ("scale_a", vp), ("scale_b", vp),
    ]


class SdpaDesc(C.Structure):
    """Mirror of favit_sdpa_t (include/favit.h)."""
    _fields_ = [
        ("q", vp), ("k", vp), ("v", vp), ("o", vp), ("lse", vp), ("dout", vp), ("dq", vp), ("dk", vp), ("dv", vp),
        ("delta", vp), ("mask", vp), ("m_sb", i64), ("m_sq", i64),
        ("q_str", i64 * 3), ("k_str", i64 * 3), ("v_str", i64 * 3), ("o_str", i64 * 3), ("do_str", i64 * 3),
        ("dq_str", i64 * 3), ("dk_str", i64 * 3), ("dv_str", i64 * 3),
        ("B", i32), ("H", i32), ("Lq", i32), ("Lk", i32), ("hd", i32), ("dtype", i32),
        ("scale", f32), ("dropout_p", f32), ("seed", u64),
    ]


_SIGS = {
    "favit_abi_version": ([], C.c_int),
    "favit_strerror": ([C.c_int], C.c_char_p),
    "favit_set_dropout_epoch": ([vp], C.c_int),
    "favit_set_health_word": ([vp], C.c_int),
    "favit_gemm": ([C.POINTER(GemmDesc), vp], C.c_int),
    "favit_gemm_last_kernel": ([], C.c_char_p),
    "favit_ln_gemm": ([C.POINTER(GemmDesc), vp, i64, vp, vp, f32, vp, vp, vp, vp], C.c_int),
    "favit_gemm_grouped_tn": ([C.POINTER(GemmDesc), i32, vp], C.c_int),
    "favit_gemm_grouped_tn_workspace": ([C.POINTER(GemmDesc), i32], C.c_int64),
    "favit_gemm_grouped_tn_ws": ([C.POINTER(GemmDesc), i32, vp, i64, vp], C.c_int),
    "favit_gemm_grouped_last_splits": ([], C.c_int),
    "favit_cast": ([vp, C.c_int, vp, C.c_int, i64, vp], C.c_int),
    "favit_fp8_amax": ([vp, C.c_int, i64, i64, i64, vp, vp], C.c_int),
    "favit_fp8_quantize": ([vp, C.c_int, i64, i64, i64, vp, i64, vp, i64, C.c_int, vp, vp, vp, vp, vp, vp], C.c_int),
    "favit_layernorm_fwd": ([vp, i64, vp, vp, vp, C.c_int, vp, vp, i64, i32, f32, vp], C.c_int),
    "favit_layernorm_bwd": ([vp, C.c_int, vp, i64, vp, vp, vp, vp, vp, i64, vp, C.c_int, vp, vp, i32, vp, vp, i32,
                             i64, i32, f32, u64, vp], C.c_int),
    "favit_layernorm_fwd_q8": ([vp, i64, vp, vp, vp, vp, vp, i64, i32, f32, vp, C.c_int, vp, vp, vp, vp, vp], C.c_int),
    "favit_layernorm_bwd_q8": ([vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, vp, vp, i32, i64, i32, f32, u64,
                                vp, C.c_int, vp, vp, vp, vp, vp], C.c_int),
    "favit_small_linear_fwd": ([vp, i64, vp, vp, vp, i32, i32, i32, vp], C.c_int),
    "favit_small_linear_bwd": ([vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, vp], C.c_int),
    "favit_reduce_rows": ([vp, i64, vp, i64, i32, i32, vp], C.c_int),
    "favit_reduce_rows_multi": ([i32, vp, vp, vp, i64, i32, vp], C.c_int),
    "favit_mhla_fold_fwd": ([vp, vp, vp, vp, vp, C.c_int, vp, vp, i32, i32, vp], C.c_int),
    "favit_mhla_fold_fwd_multi": ([i32, vp, vp, vp, vp, vp, C.c_int, vp, i32, i32, vp], C.c_int),
    "favit_mhla_fold_bwd": ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp], C.c_int),
    "favit_mhla_fold_bwd_multi": ([i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp], C.c_int),
    "favit_mhla_attn_fwd": ([vp, vp, vp, i32, i32, i32, i32, i32, C.c_int, f32, u64, vp], C.c_int),
    "favit_mhla_attn_bwd": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, C.c_int, f32, u64, vp], C.c_int),
    "favit_mhla_attn_lse_supported": ([i32, i32, i32, C.c_int], C.c_int),
    "favit_mhla_attn_fwd_lse": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, C.c_int, f32, u64, vp], C.c_int),
    "favit_mhla_attn_bwd_lse": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.c_int, f32, u64, vp], C.c_int),
    "favit_sdpa_fwd": ([C.POINTER(SdpaDesc), vp], C.c_int),
    "favit_sdpa_bwd": ([C.POINTER(SdpaDesc), vp], C.c_int),
    "favit_softmax_fwd": ([vp, vp, vp, C.c_int, vp, i64, i64, i32, i64, i32, i32, f32, u64, vp], C.c_int),
    "favit_softmax_bwd": ([vp, C.c_int, vp, vp, C.c_int, i64, i32, i32, f32, u64, vp], C.c_int),
    "favit_patchify_fwd": ([vp, vp, C.c_int, i32, i32, i32, i32, vp], C.c_int),
    "favit_patchify_bwd": ([vp, vp, i32, i32, i32, i32, vp], C.c_int),
    "favit_embed_prologue_fwd": ([vp, vp, vp, vp, i32, i32, i32, vp], C.c_int),
    "favit_embed_prologue_bwd": ([vp, vp, C.c_int, vp, vp, i32, i32, i32, vp], C.c_int),
    "favit_dropout": ([vp, vp, C.c_int, i64, f32, u64, vp], C.c_int),
    "favit_sppp_map_patches": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, vp], C.c_int),
    "favit_sppp_pool_fwd": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], C.c_int),
    "favit_sppp_pool_bwd": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], C.c_int),
    "favit_sppp_centroids": ([vp, vp, i32, i32, i32, vp], C.c_int),
    "favit_sppp_posenc_fwd": ([vp, vp, vp, i32, i32, i32, i32, vp], C.c_int),
    "favit_image_transform": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp], C.c_int),
    "favit_slic_features_workspace": ([i32, i32, i32], C.c_int64),
    "favit_slic_features": ([vp, vp, i32, i32, i32, f32, i32, vp, vp], C.c_int),
    "favit_slic_cluster_workspace": ([i32, i32], C.c_int64),
    "favit_slic_cluster": ([vp, vp, vp, i32, i32, i32, i32, i32, i64, i32, vp, vp], C.c_int),
    "favit_slic_connect": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, vp], C.c_int),
    "favit_cross_entropy": ([vp, vp, vp, vp, i32, i32, f32, vp], C.c_int),
    "favit_adamw": ([vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, f32, f32, vp], C.c_int),
}

_lib = None


class FavitLibraryError(RuntimeError):
    pass


def declared_symbols():
    """Entry points declared in include/favit.h (used by the symbol-export test)."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(favit_[a-z0-9_]+)\s*\(", txt)))


def lib():
    """Load libfavit.so (once).  Raises FavitLibraryError if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FavitLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C focused-attention-vit_amd/csrc`.  There is no CPU / PyTorch fallback.")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise FavitLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        _lib = l
    return _lib


def check(code: int, what: str):
    if code != 0:
        msg = lib().favit_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
