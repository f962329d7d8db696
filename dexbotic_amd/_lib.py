"""ctypes binding of libdexbotic_amd.so (C ABI declared in include/dexbotic_amd.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this module
raises at import; every kernel call checks the status code and raises ``DxaError`` with
``dxa_last_error()``.  Build the library with ``python -m dexbotic_amd.build`` (done by
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

# torch first: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7; whichever copy of that soname is
# mapped first serves the whole process, and device pointers / streams handed to our kernels belong to
# torch's runtime.  Loading libdexbotic_amd.so before torch would bind everything to /opt/rocm's copy,
# which then fails to see the device torch initialised ("no ROCm-capable device").
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DXA_LIB") or os.path.join(_HERE, "libdexbotic_amd.so")   # DXA_LIB: kernel-tuning builds

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH, ACT_QUICK_GELU, ACT_SILU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3, 4, 5, 6
NT, NN, TN = 0, 1, 2
FILTER_BICUBIC = 3
FUSE_NONE, FUSE_SWIGLU = 0, 1

_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
_int = C.c_int


class DxaError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("layout", _i32), ("in_dtype", _i32), ("out_dtype", _i32), ("act", _i32),
        ("M", _i64), ("N", _i64), ("K", _i64),
        ("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64),
        ("bias", _vp), ("residual", _vp), ("ldr", _i64), ("aux_out", _vp), ("mulgrad", _vp), ("ldg", _i64),
        ("alpha", _f32), ("accumulate", _i32), ("nb", _i32 * 3),
        ("sA", _i64 * 3), ("sB", _i64 * 3), ("sC", _i64 * 3), ("sR", _i64 * 3), ("sG", _i64 * 3),
        ("epi_f32", _i32), ("mirror", _vp), ("sumsq", _vp),
        ("A2", _vp), ("B2", _vp), ("K2", _i64),
        ("fuse", _i32), ("ld_aux", _i64),
    ]


class Split3Op(C.Structure):
    _fields_ = [("src", _vp), ("ld", _i64), ("dst", _vp), ("rows", _i64), ("cols", _i64), ("pad", _i64), ("side", _i32), ("transposed", _i32)]


class AttnDesc(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("B", _i32), ("Hq", _i32), ("Hkv", _i32), ("Sq", _i32), ("Sk", _i32), ("D", _i32),
        ("causal", _i32), ("scale", _f32),
        ("q", _vp), ("q_sb", _i64), ("q_sh", _i64), ("q_ss", _i64),
        ("k", _vp), ("k_sb", _i64), ("k_sh", _i64), ("k_ss", _i64),
        ("v", _vp), ("v_sb", _i64), ("v_sh", _i64), ("v_ss", _i64),
        ("o", _vp), ("o_sb", _i64), ("o_sh", _i64), ("o_ss", _i64),
        ("lse", _vp), ("kv_start", _vp), ("kv_end", _vp),
        ("d_o", _vp), ("do_sb", _i64), ("do_sh", _i64), ("do_ss", _i64),
        ("dq", _vp), ("dq_sb", _i64), ("dq_sh", _i64), ("dq_ss", _i64),
        ("dk", _vp), ("dk_sb", _i64), ("dk_sh", _i64), ("dk_ss", _i64),
        ("dv", _vp), ("dv_sb", _i64), ("dv_sh", _i64), ("dv_ss", _i64),
        ("force_generic", _i32),
        ("q_limit", _vp), ("key_valid", _vp), ("drop_mask", _vp),
    ]


class AdamWDesc(C.Structure):
    _fields_ = [
        ("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("shadow", _vp),
        ("chunk_start", _vp), ("chunk_len", _vp), ("chunk_grp", _vp), ("n_chunks", _i32),
        ("lr", _f32 * 8), ("wd", _f32 * 8),
        ("beta1", _f32), ("beta2", _f32), ("eps", _f32), ("bc1", _f32), ("bc2", _f32),
        ("clip_coef", _vp), ("g_dtype", _i32), ("chunk_state", _vp), ("chunk_mv_start", _vp),
    ]


class ImageDesc(C.Structure):
    _fields_ = [
        ("src", _vp), ("n", _i32), ("h", _i32), ("w", _i32), ("pad", _i32), ("bg", C.c_ubyte * 4),
        ("res_h", _i32), ("res_w", _i32), ("crop_top", _i32), ("crop_left", _i32), ("out_h", _i32), ("out_w", _i32),
        ("row0", _i32), ("rows", _i32), ("hb", _vp), ("hk", _vp), ("hks", _i32), ("vb", _vp), ("vk", _vp), ("vks", _i32),
        ("tmp", _vp), ("out", _vp), ("out_dtype", _i32), ("out_u8", _vp), ("rescale", C.c_double),
        ("mean", _f32 * 3), ("std", _f32 * 3),
    ]


class DecodeDesc(C.Structure):
    _fields_ = [
        ("layers", _vp), ("x_in", _vp), ("out", _vp), ("final_norm_w", _vp), ("cos_row", _vp), ("sin_row", _vp),
        ("workspace", _vp), ("workspace_bytes", _sz),
        ("n_layers", _i32), ("d", _i32), ("Hq", _i32), ("Hkv", _i32), ("D", _i32), ("F", _i32), ("slot", _i32), ("kv_lo", _i32),
        ("max_len", _i32), ("eps", _f32),
    ]


# name -> (restype, argtypes); must list EVERY symbol declared in include/dexbotic_amd.h
SIGNATURES = {
    "dxa_last_error": (C.c_char_p, []),
    "dxa_version": (_int, []),
    "dxa_gemm": (_int, [C.POINTER(GemmDesc), _vp]),
    "dxa_split3": (_int, [_vp, _i64, _vp, _i64, _i64, _int, _vp]),
    "dxa_split3_t": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _int, _vp]),
    "dxa_split3_pair": (_int, [C.POINTER(Split3Op), C.POINTER(Split3Op), _vp]),
    "dxa_rmsnorm_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _int, _int, _vp]),
    "dxa_rmsnorm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _vp]),
    "dxa_layernorm_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _int, _int, _vp]),
    "dxa_layernorm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _vp]),
    "dxa_norm_bwd_blocks": (_int, [_i64]),
    "dxa_colsum": (_int, [_vp, _i64, _vp, _i64, _i64, _int, _int, _vp, _sz, _vp]),
    "dxa_rope_split": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "dxa_rope_merge": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "dxa_rope_split_at": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "dxa_rope_merge_at": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "dxa_attn_fwd": (_int, [C.POINTER(AttnDesc), _vp]),
    "dxa_attn_fwd_workspace": (_sz, [C.POINTER(AttnDesc)]),
    "dxa_attn_fwd_ws": (_int, [C.POINTER(AttnDesc), _vp, _sz, _vp]),
    "dxa_attn_bwd_workspace": (_sz, [C.POINTER(AttnDesc)]),
    "dxa_attn_bwd": (_int, [C.POINTER(AttnDesc), _vp, _sz, _vp]),
    "dxa_swiglu_fwd": (_int, [_vp, _vp, _i64, _i64, _int, _vp]),
    "dxa_swiglu_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "dxa_axpby": (_int, [_vp, _vp, _vp, _i64, _f32, _f32, _int, _vp]),
    "dxa_mul": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "dxa_mul_rows": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "dxa_glu_fwd": (_int, [_vp, _vp, _i64, _i64, _int, _int, _vp]),
    "dxa_glu_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _vp]),
    "dxa_act_fwd": (_int, [_vp, _vp, _i64, _int, _int, _vp]),
    "dxa_act_bwd": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "dxa_add": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "dxa_cast": (_int, [_vp, _vp, _i64, _int, _int, _vp]),
    "dxa_copy2d": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _int, _vp]),
    "dxa_transpose": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _vp]),
    "dxa_permute_bshd": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "dxa_splice_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "dxa_splice_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "dxa_zero_rows": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "dxa_gather_rows": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _vp]),
    "dxa_scatter_rows": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _int, _int, _vp]),
    "dxa_im2col": (_int, [_vp, _vp, _int, _int, _int, _int, _i64, _int, _int, _vp]),
    "dxa_vit_embed_fwd": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "dxa_vit_embed_bwd": (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    "dxa_qsample": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "dxa_timestep_embedding": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "dxa_dit_assemble_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "dxa_dit_assemble_bwd": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    "dxa_token_drop": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "dxa_token_drop_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "dxa_dropout_mask": (_int, [_vp, _i64, _f32, C.c_uint64, C.c_uint64, _int, _vp]),
    "dxa_bank_consolidate": (_int, [_vp, _vp, _int, _i64, _i64, _int, _int, _vp, _vp]),
    "dxa_add_rows": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _f32, _int, _vp]),
    "dxa_token_sum": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _int, _vp]),
    "dxa_mse_loss": (_int, [_vp, _vp, _vp, _vp, _i64, _f32, _vp]),
    "dxa_mse_loss_rows": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _vp]),
    "dxa_ddim_step": (_int, [_vp, _vp, _i64, _i64, _int, _f32, _f32, _f32, _f32, _vp]),
    "dxa_adamw": (_int, [C.POINTER(AdamWDesc), _vp]),
    "dxa_sumsq": (_int, [_vp, _i64, _int, _vp, _vp, _int, _vp]),
    "dxa_sumsq_ranges": (_int, [_vp, _int, _vp, _vp, _int, _vp, _vp, _int, _vp]),
    "dxa_sum_f32": (_int, [_vp, _i64, _vp, _int, _vp]),
    "dxa_gemm_sumsq_slots": (_i64, [_i64, _i64]),
    "dxa_cross_entropy_fwd": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "dxa_cross_entropy_bwd": (_int, [_vp, _i64, _vp, _vp, _vp, _f32, _vp, _i64, _i64, _i64, _i64, _int, _vp]),
    "dxa_argmax_rows": (_int, [_vp, _i64, _vp, _i64, _i64, _int, _vp]),
    "dxa_resample_ksize": (_int, [_int, _int]),
    "dxa_resample_coeffs": (_int, [_int, _int, _int, _vp, _vp]),
    "dxa_image_preprocess": (_int, [C.POINTER(ImageDesc), _vp]),
    "dxa_dit_blocks_workspace": (_sz, [_int, _int, _int]),
    "dxa_dit_blocks_fwd": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "dxa_dit_blocks_status": (_int, [_vp, _vp]),
    "dxa_dit_sample_workspace": (_sz, [_int, _int, _int]),
    "dxa_dit_sample_fwd": (_int, [_vp] * 9 + [_int, _int, _int, _int, _f32, _vp, _int, _int, _int, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "dxa_dit_bf16_pack_bytes": (_sz, [_int, _int, _int]),
    "dxa_dit_bf16_pack": (_int, [_vp, _int, _int, _int, _vp, _sz, _vp, _vp]),
    "dxa_dit_bf16_pack_per_bytes": (_sz, [_int, _int, _int]),
    "dxa_dit_bf16_pack_per": (_int, [_vp, _int, _int, _int, _vp, _sz, _vp, _vp]),
    "dxa_dit_sample_bf16_per_fwd": (_int, [_vp] * 9 + [_int, _int, _int, _int, _f32, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "dxa_dit_sample_bf16_workspace": (_sz, [_int, _int, _int]),
    "dxa_dit_sample_bf16_fwd": (_int, [_vp] * 9 + [_int, _int, _int, _int, _f32, _vp, _int, _int, _int, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "dxa_decode_step_workspace": (_sz, [_int, _int, _int, _int, _int]),
    "dxa_decode_step": (_int, [C.POINTER(DecodeDesc), _vp]),
    "dxa_decode_status": (_int, [_vp, _vp]),
    "dxa_clip_coef": (_int, [_vp, _f32, _vp, _vp, _vp]),
    "dxa_clip_coef_scaled": (_int, [_vp, _f32, _f32, _vp, _vp, _vp]),
    "dxa_scale": (_int, [_vp, _i64, _f32, _vp]),
    "dxa_scale_dev": (_int, [_vp, _i64, _vp, _vp]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"dexbotic_amd: native library {LIB_PATH} not found. Build it with "
            "`python -m dexbotic_amd.build` (hipcc --offload-arch=gfx950); there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"dexbotic_amd: symbol {name} missing from {LIB_PATH} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error() -> str:
    return (lib.dxa_last_error() or b"").decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise DxaError(f"{what}: status {rc}: {last_error()}")
