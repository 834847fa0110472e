"""ctypes loader for libcream_amd.so — the C-ABI HIP library (include/cream_amd.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing or a
symbol cannot be resolved this module raises immediately (`CreamLibraryError`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcream_amd.so")

# error codes of include/cream_amd.h
CREAM_OK = 0
_ERRORS = {
    -1: "CREAM_ERR_BAD_ARG (null pointer, negative size or unsupported stride)",
    -2: "CREAM_ERR_BAD_DTYPE (element type not supported by this entry point)",
    -3: "CREAM_ERR_LAUNCH (HIP kernel launch failed)",
    -4: "CREAM_ERR_TOO_LARGE (shape exceeds what the kernel family supports)",
}

# dtype enum of include/cream_amd.h
F32, F16, BF16, F64 = 0, 1, 2, 3


class CreamLibraryError(RuntimeError):
    pass


_c = ctypes
_vp, _i, _i64, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

# name -> (restype, argtypes).  Must list every symbol include/cream_amd.h declares;
# tests/test_cabi.py checks header and table against each other.
SIGNATURES = {
    "cream_version": (_c.c_char_p, []),
    "cream_build_info": (_c.c_char_p, []),
    "cream_rpe_index_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _vp]),
    "cream_rpe_index_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cream_rpe_index_fwd_host": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "cream_rpe_index_bwd_host": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "cream_im2patch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cream_stem_assemble": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "cream_stem_bwd_chunks": (_i, [_i]),
    "cream_stem_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "cream_tail_chunks": (_i, [_i]),
    "cream_tail_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "cream_tail_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "cream_irpe_padded_len": (_i, [_i]),
    "cream_irpe_bucket_bytes": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "cream_irpe_attn_fwd": (_i, [_vp, _vp]),
    "cream_irpe_attn_bwd": (_i, [_vp, _vp]),
    "cream_irpe_table_grad": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i, _i, _i, _f, _vp]),
    "cream_attn_rpe2d_padded_len": (_i, [_i]),
    "cream_attn_rpe2d_dtab_parts": (_i, [_i, _i]),
    "cream_attn_rpe2d_bwd_mode": (_i, [_i]),
    "cream_attn_rpe2d_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i,
                                  _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cream_attn_rpe2d_bwd": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                  _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cream_attn_rpe2d_fwd_img": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i, _vp,
                                      _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cream_attn_rpe2d_bwd_img": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                      _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cream_attn_rpe2d_fwd_drop": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i, _vp,
                                       _i, _i, _i, _i, _i, _i, _f, _f, _c.c_uint32, _i, _vp]),
    "cream_attn_rpe2d_bwd_drop": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                       _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _c.c_uint32, _i, _vp]),
    "cream_attn_rpe2d_table_image_bytes": (_i64, []),
    "cream_attn_rpe2d_table_images": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "cream_attn_rpe2d_fwd_mode": (_i, [_i]),
    "cream_ln_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "cream_ln_partials": (_i, []),
    "cream_ln_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "cream_gelu_fwd": (_i, [_vp, _vp, _i64, _vp]),
    "cream_gelu_bwd": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "cream_residual_add": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "cream_scale_cast": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "cream_colsum_slabs": (_i, [_i]),
    "cream_colsum": (_i, [_vp, _vp, _i, _i, _vp]),
    "cream_add_ln_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _vp]),
    "cream_block_wgrad_bf16": (_i, [_i]),
    "cream_colsum128_slabs": (_i, [_i]),
    "cream_colsum128": (_i, [_vp, _vp, _i, _i, _vp]),
    "cream_gelu_bwd_colsum": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "cream_scale_cast_colsum": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "cream_grad_finalize": (_i, [_vp, _i, _vp]),
    "cream_gemm_rows_per_colsum_slab": (_i, []),
    "cream_gemm_nt256": (_i, [_i]),
    "cream_gemm_nt8": (_i, [_i]),
    "cream_linear_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    "cream_linear_fwd_seg": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i, _i64, _vp]),
    "cream_linear_gelu_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    "cream_linear_gelu_fwd_pad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _vp]),
    "cream_linear_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    "cream_linear_dgrad_seg": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i, _i64, _vp]),
    "cream_linear_dgrad_mul": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    "cream_linear_wgrad_splits": (_i, [_i, _i, _i]),
    "cream_linear_wgrad_splits_bf16": (_i, [_i, _i, _i]),
    "cream_gemm_tn8": (_i, [_i]),
    "cream_gemm_nthalf": (_i, [_i]),
    "cream_gemm_ntopt": (_i, [_i]),
    "cream_gemm_stagger": (_i, [_i]),
    "cream_block_layout_epoch": (_i, []),
    "cream_cu_reserve": (_i, [_i]),
    "cream_mixup_cutmix": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _f, _vp]),
    "cream_image_batch_plan": (_i64, [_vp, _i, _i, _i]),
    "cream_image_batch_transform": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    "cream_cu_count": (_i, []),
    "cream_linear_wgrad_parts": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "cream_linear_wgrad_parts_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "cream_param_job_tiles": (_i, [_i, _i]),
    "cream_adamw_step": (_i, [_vp, _vp, _i, _i, _i, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _i64, _vp]),
    "cream_soft_ce": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "cream_linear_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i, _i, _vp]),
    "cream_bmm_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "cream_linear_f32_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i, _i, _vp]),
    "cream_linear_f32_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i, _i, _vp]),
    "cream_ln_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "cream_ln_f32_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "cream_block_prof_enable": (_i, [_i]),
    "cream_block_prof_kinds": (_i, []),
    "cream_block_prof_name": (_c.c_char_p, [_i]),
    "cream_block_prof_collect": (_i, [_vp, _vp, _vp, _vp]),
    "cream_slices_copy": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "cream_block_fwd_workspace": (_i64, [_vp, _vp, _vp, _vp]),
    "cream_block_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cream_block_bwd_workspace": (_i64, [_vp, _vp, _vp, _vp]),
    "cream_block_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _i, _vp, _vp]),
}

MAX_GRAD_JOBS = 24


class GradJob(ctypes.Structure):
    """struct cream_grad_job of include/cream_amd.h."""
    _fields_ = [("dst", _vp), ("src", _vp), ("ld", _i64), ("pstride", _i64),
                ("nparts", _c.c_int32), ("rows", _c.c_int32), ("cols", _c.c_int32),
                ("interleave", _c.c_int32), ("src_bf16", _c.c_int32), ("overwrite", _c.c_int32)]

class ImageDesc(ctypes.Structure):
    """struct cream_image_desc of include/cream_amd.h."""
    _fields_ = ([("offset", _i64)] +
                [(n, _c.c_int32) for n in ("height", "width", "row_stride", "box_top", "box_left", "box_h", "box_w", "resized_h",
                                           "resized_w", "win_top", "win_left", "flip", "row0", "nrows")] +
                [("tmp_off", _i64)] +
                [(n, _c.c_int32) for n in ("erase_top", "erase_left", "erase_h", "erase_w")] +
                [("erase_seed", _c.c_uint32), ("reserved", _c.c_int32)])


class SliceJob(ctypes.Structure):
    """struct cream_slice_job of include/cream_amd.h."""
    _fields_ = [("full", _vp), ("ld", _i64), ("packed_off", _i64), ("rows", _c.c_int32), ("cols", _c.c_int32)]


class BlockDesc(ctypes.Structure):
    """struct cream_block_desc of include/cream_amd.h."""
    _fields_ = ([(n, _c.c_int32) for n in ("B", "N", "E", "H", "F", "gh", "gw", "mr", "F_valid", "inference")] +
                [(n, _f) for n in ("eps1", "eps2", "attn_scale", "reserved_f")] +
                [(n, _vp) for n in ("wqkv", "wqkv_t", "bqkv", "wproj", "wproj_t", "bproj", "w1", "w1_t", "b1",
                                    "w2", "w2_t", "b2")] +
                [(n, _i64) for n in ("ld_qkv", "ld_qkv_t", "seg_qkv", "seg_qkv_t", "ld_proj", "ld_proj_t", "ld_w1",
                                     "ld_w1_t", "ld_w2", "ld_w2_t")] +
                [(n, _vp) for n in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "tkv", "tkh", "tvv", "tvh")] +
                [("ldt", _i64), ("timg", _vp)])


class IrpeAttnDesc(ctypes.Structure):
    """struct cream_irpe_attn_desc of include/cream_amd.h."""
    _fields_ = ([(n, _vp) for n in ("q", "k", "v")] + [(n, _i64) for n in ("sb", "sn", "sh")] +
                [(n, _vp) for n in ("out", "lse", "sv", "wq", "wk", "wv")] +
                [(n, _i64) for n in ("wq_hs", "wk_hs", "wv_hs")] +
                [(n, _vp) for n in ("idq", "idk", "idv", "idq_t", "idk_t", "idv_t")] +
                [(n, _c.c_int32) for n in ("B", "H", "L", "NP", "nb")] + [("scale", _f)] +
                [(n, _vp) for n in ("dout", "dq", "dk", "dv")] + [(n, _i64) for n in ("dsb", "dsn", "dsh")] +
                [(n, _vp) for n in ("delta", "lkg", "gg", "dlk", "dlq", "bq", "bk")] + [(n, _i64) for n in ("bq_hs", "bk_hs")] +
                [("causal", _c.c_int32), ("reserved", _c.c_int32), ("dropout_p", _c.c_float), ("dropout_seed", _c.c_uint32)])


class ParamJob(ctypes.Structure):
    """struct cream_param_job of include/cream_amd.h."""
    _fields_ = ([(n, _vp) for n in ("p", "g", "m", "v", "mir", "mir_t")] +
                [(n, _i64) for n in ("ld", "ld_mir", "ld_mir_t", "seg_stride", "seg_stride_t")] +
                [(n, _c.c_int32) for n in ("rows", "cols", "deinterleave")] + [("weight_decay", _f)])


class BlockGrads(ctypes.Structure):
    """struct cream_block_grads of include/cream_amd.h."""
    _fields_ = ([(n, _vp) for n in ("wqkv", "bqkv", "wproj", "bproj", "w1", "b1", "w2", "b2", "ln1_g", "ln1_b", "ln2_g",
                                    "ln2_b", "tkv", "tkh", "tvv", "tvh")] +
                [(n, _i64) for n in ("ld_qkv", "ld_proj", "ld_w1", "ld_w2", "ldt")])


_lib = None


def load():
    """Load the library once and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CreamLibraryError(
            f"cream_amd: {LIB_PATH} is missing — build it with `python -m cream_amd.build` "
            "(or __graft_entry__.build()).  There is no fallback path.")
    # PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  It must be
    # mapped FIRST so that this library binds to the runtime that owns torch's streams
    # and allocations; two HIP runtimes in one process make every launch fail.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise CreamLibraryError(f"cream_amd: cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CreamLibraryError(f"cream_amd: symbol {name} missing from {LIB_PATH}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != CREAM_OK:
        raise RuntimeError(f"cream_amd: {what} failed with {_ERRORS.get(code, code)}")


def version():
    return load().cream_version().decode()


def build_info():
    return load().cream_build_info().decode()
