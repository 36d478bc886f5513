"""ctypes binding of liblbmk.so (the C-ABI declared in include/lbmk.h).

The library is built in-tree by ``make -C lbm_amd/csrc`` (or ``__graft_entry__.build()``).
There is no fallback: importing this module without the shared library raises.
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64.so.  It must be the first HIP runtime in the process:
# liblbmk.so then binds to it (same SONAME) and shares torch's device context and streams.
# Loading liblbmk.so first would pull in /opt/rocm's copy as a second runtime, whose launches fail
# with "no ROCm-capable device is detected" once torch has initialised its own.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblbmk.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP kernels first "
        "(`make -C lbm_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "lbm_amd has no CPU or eager-PyTorch fallback."
    )

lib = C.CDLL(LIB_PATH)

LBMK_OK = 0
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_RELU = 0, 1, 2, 3
DT_BF16, DT_F32 = 0, 1

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class IgemmDesc(C.Structure):
    _fields_ = [
        ("a", vp), ("w", vp), ("out", vp), ("bias", vp), ("rowvec", vp), ("residual", vp),
        ("M", i32), ("N", i32), ("K", i32), ("Kp", i32),
        ("Cin", i32), ("a_cs", i32), ("a_coff", i32),
        ("IH", i32), ("IW", i32), ("OH", i32), ("OW", i32),
        ("KH", i32), ("KW", i32), ("stride", i32), ("pad_t", i32), ("pad_l", i32),
        ("up2", i32),
        ("ldo", i32), ("ldr", i32), ("rowvec_ld", i32),
        ("act", i32), ("out_f32", i32), ("bias_per_m", i32),
        ("alpha", f32),
        ("batch", i32),
        ("sa", i64), ("sw", i64), ("so", i64), ("sr", i64),
        ("tile", i32), ("k_order", i32),
        ("gn_partial", vp), ("gn_groups", i32),
        ("batch_inner", i32), ("sa2", i64), ("sw2", i64), ("so2", i64),
        ("in_f32", i32),
        ("out_up2", i32), ("gn_nslab", i32), ("gn_strip_off", i32),
        ("smx_mode", i32), ("smx_max", vp), ("smx_part", vp),
        ("f32_chunks", i32), ("f32_part", vp),
    ]


# name -> (restype, argtypes); mirrors include/lbmk.h one to one
SIGNATURES = {
    "lbmk_version": (i32, []),
    "lbmk_last_error": (C.c_char_p, []),
    "lbmk_igemm": (i32, [C.POINTER(IgemmDesc), vp]),
    "lbmk_igemm_tiles": (i32, [C.POINTER(i32), C.POINTER(i32), i32]),
    "lbmk_igemm_smx_parts": (i32, [i32]),
    "lbmk_igemm_f32_chunks": (i32, [i32, i32, i32]),
    "lbmk_igemm_f32_split_pays": (i32, [i32, i32, i32, i32, i32]),
    "lbmk_igemm_gn_partial_floats": (i64, [i32, i32, i32, i32, i32, i32]),
    "lbmk_groupnorm_from_partials": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, f32, i32, vp]),
    "lbmk_groupnorm_scratch_floats": (i64, [i32, i32, i32, i32]),
    "lbmk_groupnorm": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "lbmk_groupnorm_static_max_floats": (i64, []),
    "lbmk_groupnorm_static": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "lbmk_layernorm": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "lbmk_flash_attn_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                  i64, i64, i64, i64, f32, vp, vp]),
    "lbmk_flash_attn_bwd": (i32, [vp] * 13 + [i32] * 9 + [i64] * 6 + [f32, vp]),
    "lbmk_softmax_rows": (i32, [vp, vp, i64, i32, i64, i64, vp]),
    "lbmk_bridge_interp": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, f32, i32, i32, vp]),
    "lbmk_euler_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, i32, vp]),
    "lbmk_latent_loss": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp]),
    "lbmk_axpy_rows": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "lbmk_nchw_to_nhwc": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "lbmk_nhwc_to_nchw": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "lbmk_copy_channels": (i32, [vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "lbmk_posterior_sample": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, f32, i32, vp]),
    "lbmk_timestep_embedding": (i32, [vp, vp, i32, i32, vp]),
    "lbmk_silu": (i32, [vp, vp, i64, vp]),
    "lbmk_scale_shift": (i32, [vp, vp, i64, f32, f32, vp]),
    "lbmk_tile_accumulate": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                   i32, i32, vp]),
    "lbmk_tile_finalize": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "lbmk_transpose": (i32, [vp, vp, i64, i32, i64, i64, i32, i64, i64, vp]),
    "lbmk_im2col_t": (i32, [vp, vp] + [i32] * 14 + [i64, vp]),
    "lbmk_col2im": (i32, [vp, vp] + [i32] * 12 + [i64, i32, vp]),
    "lbmk_colsum_scratch_floats": (i64, [i64, i64, i32]),
    "lbmk_colsum": (i32, [vp, vp, vp, i64, i64, i32, i64, i32, vp]),
    "lbmk_sum_partials": (i32, [vp, vp, i64, i32, i32, vp]),
    "lbmk_splitk_finish": (i32, [vp, i32, i64, vp, vp, i32, vp, i32, i32, i32, vp]),
    "lbmk_gemm_tn": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lbmk_gemm_nn": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lbmk_pack_conv_weight": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lbmk_lpips_prep": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "lbmk_lpips_prep_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "lbmk_relu_bwd": (i32, [vp, vp, vp, i64, vp]),
    "lbmk_maxpool2": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "lbmk_maxpool2_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "lbmk_lpips_head_partial_floats": (i64, [i32, i32, i32]),
    "lbmk_lpips_head": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "lbmk_lpips_head_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "lbmk_groupnorm_bwd_scratch_floats": (i64, [i32, i32, i32, i32]),
    "lbmk_groupnorm_bwd": (i32, [vp] * 9 + [i32] * 5 + [vp]),
    "lbmk_layernorm_bwd_scratch_floats": (i64, [i32, i32]),
    "lbmk_layernorm_bwd": (i32, [vp] * 6 + [i32, i32, f32, vp]),
    "lbmk_layernorm_bwd_ex": (i32, [vp] * 6 + [i32, vp, vp, i32, i32, f32, vp]),
    "lbmk_groupnorm_bwd_ex": (i32, [vp] * 8 + [i32, vp, vp] + [i32] * 5 + [vp]),
    "lbmk_silu_bwd": (i32, [vp, vp, vp, i64, vp]),
    "lbmk_geglu_fwd": (i32, [vp, vp, i64, i32, vp]),
    "lbmk_geglu_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "lbmk_softmax_bwd_rows": (i32, [vp, vp, vp, i64, i32, f32, vp]),
    "lbmk_latent_loss_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, vp]),
    "lbmk_pixel_loss": (i32, [vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]),
    "lbmk_pixel_loss_bwd": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]),
    "lbmk_groupnorm_f32_scratch_floats": (i64, [i32, i32]),
    "lbmk_groupnorm_f32": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "lbmk_layernorm_f32": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "lbmk_softmax_rows_f32": (i32, [vp, vp, i64, i32, i64, i64, vp]),
    "lbmk_geglu_f32": (i32, [vp, vp, i64, i32, vp]),
    "lbmk_nchw_to_nhwc_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "lbmk_nhwc_to_nchw_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "lbmk_copy_channels_f32": (i32, [vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "lbmk_posterior_sample_f32": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "lbmk_timestep_embedding_f32": (i32, [vp, vp, i32, i32, vp]),
    "lbmk_silu_f32": (i32, [vp, vp, i64, vp]),
    "lbmk_scale_shift_f32": (i32, [vp, vp, i64, f32, f32, vp]),
    "lbmk_tile_accumulate_f32": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                       i32, i32, vp]),
    "lbmk_adamw": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
}

# present only in diagnostic builds of the library (make DIAG=1)
if hasattr(lib, "lbmk_debug_igemm_times"):
    SIGNATURES["lbmk_debug_igemm_times"] = (i32, [vp, i32])

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here = header / library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


class LbmkError(RuntimeError):
    pass


def check(status, what=""):
    if status != LBMK_OK:
        msg = lib.lbmk_last_error().decode("utf-8", "replace")
        raise LbmkError(f"{what or 'lbmk'} failed (status {status}): {msg}")
