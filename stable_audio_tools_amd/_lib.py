"""ctypes binding of the C-ABI in include/sat_amd.h.

The product path loads exactly one library: ``csrc/libsat_amd.so`` (hipcc, gfx950).  If it is
missing or cannot be loaded this module raises — there is NO CPU or PyTorch fallback for the
kernels.  (The CPU test-suite binds the same signatures onto the host-side *simulator* build of the
kernel sources, tests/emu/libsat_emu.so, through :func:`bind`; :func:`load` refuses that library.)
"""
import ctypes
import os

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_longlong
_F = ctypes.c_float

# name -> (restype, argtypes).  Must match include/sat_amd.h (tests/test_abi.py checks both ways).
SIGNATURES = {
    "sat_abi_version": (_I, []),
    "sat_is_simulator": (_I, []),
    "sat_last_error": (ctypes.c_char_p, []),
    # conv1d.hip
    "sat_conv1d": (_I, [_P] * 12 + [_I] * 10 + [_P]),
    "sat_conv1d_partial_rows": (_I, [_I, _I]),
    # conv1d_bf16x3.hip
    "sat_conv1d_bf16x3": (_I, [_P] * 13 + [_I] * 10 + [_P]),
    "sat_conv1d_bf16x3_emit": (_I, [_P] * 13 + [_I] * 10 + [_P] * 4 + [_I, _P]),
    "sat_convtr1d_bf16x3": (_I, [_P] * 13 + [_I] * 9 + [_P]),
    "sat_conv1d_bf16x3_partial_rows": (_I, [_I] * 4),
    "sat_rows_pack": (_I, [_P, _P] + [_I] * 10 + [_P]),
    "sat_rows_pack_bwd": (_I, [_P, _P] + [_I] * 10 + [_P]),
    "sat_rows_unpack": (_I, [_P, _P] + [_I] * 6 + [_F, _P]),
    "sat_rows_unpack_bwd": (_I, [_P, _P, _P] + [_I] * 6 + [_F, _P]),
    # disc_conv.hip
    "sat_disc_geom": (_I, [_I, _I, _P, _P, _P, _P]),
    "sat_disc_planes": (_I, [_P] * 7 + [_I] * 5 + [_F, _P]),
    "sat_disc_l1_blocks": (_I, []),
    "sat_disc_l1_sum": (_I, [_P, _P, _P, _P, _L, _P]),
    "sat_disc_pack_size": (_L, [_I] * 5),
    "sat_disc_pack_weights": (_I, [_P] * 2 + [_I] * 5 + [_P]),
    "sat_disc_conv": (_I, [_P] * 7 + [_I] * 8 + [_F, _P, _F, _P]),
    "sat_disc_wgrad_nsplit": (_I, [_I] * 6),
    "sat_disc_wgrad": (_I, [_P] * 3 + [_I] * 8 + [_P]),
    "sat_conv1d_k7_plane_rows": (_I, [_I] * 3),
    "sat_conv1d_k7_planes": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "sat_conv1d_bf16x3_planesq": (_I, [_P, _P, _I] + [_P] * 10 + [_I] * 10 + [_P]),
    "sat_pack_weights_k7q": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "sat_residual_unit_fwd": (_I, [_P, _P, _I] + [_P] * 11 + [_I] * 6 + [_P] * 4 + [_I, _P]),
    "sat_pack_weights_k7q_size": (_L, [_I, _I, _I, _I]),
    "sat_convtr1d_bf16x3_partial_rows": (_I, [_I] * 4),
    "sat_pack_weights_bf16x3": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sat_pack_weights_bf16x3_size": (_L, [_I, _I, _I, _I, _I]),
    "sat_snake_consts": (_I, [_P, _P, _P, _P, _I, _P]),
    # conv_wgrad_bf16x3.hip
    "sat_conv_wgrad7_bf16x3": (_I, [_P] * 5 + [_L] * 3 + [_I] * 6 + [_P, _P]),
    "sat_conv_wgrad7_bf16x3_nsplit": (_I, [_I] * 4),
    "sat_conv_wgrad7_bf16x3_fuses_rowsum": (_I, [_I] * 4),
    "sat_conv_wgrad_bf16x3": (_I, [_P] * 4 + [_I, _P] + [_L] * 3 + [_I] * 8 + [_P, _P]),
    "sat_conv_wgrad_bf16x3_nsplit": (_I, [_I] * 6),
    # convtr1d.hip
    "sat_convtr1d": (_I, [_P] * 12 + [_I] * 9 + [_P]),
    "sat_convtr1d_partial_rows": (_I, [_I, _I, _I, _I]),
    # conv_wgrad.hip
    "sat_conv_wgrad": (_I, [_P, _P, _P, _P, _I, _P, _L, _L, _L] + [_I] * 9 + [_P]),
    "sat_conv_wgrad_nsplit": (_I, [_I] * 7),
    "sat_reduce_splits": (_I, [_P, _P, _L, _I, _F, _I, _P]),
    "sat_ru_k1_bwd_nsplit": (_I, [_I] * 3),
    "sat_ru_k1_pack": (_I, [_P, _P, _P, _I, _P]),
    "sat_ru_k1_bwd": (_I, [_P] * 9 + [_I, _P, _P, _I, _I, _I, _P]),
    "sat_rowsum": (_I, [_P, _P, _I, _I, _I, _P]),
    "sat_rowsum_nsplit": (_I, [_I]),
    # elementwise.hip
    "sat_wn_fold": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "sat_wn_grad": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sat_wn_grad_splits": (_I, [_P, _I, _L, _L, _L, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P]),
    "sat_pack_weights": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sat_vae_nblocks": (_I, [_L]),
    "sat_vae_sample_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "sat_vae_sample_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sat_multi_copy": (_I, [_P, _I, _P]),
    "sat_adamw_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P, _F, _P]),
    "sat_adamw_step_dev": (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P, _P]),
    # comm.hip
    "sat_allreduce_available": (_I, []),
    "sat_allreduce_unique_id": (_I, [_P]),
    "sat_allreduce_init": (_I, [_P, _I, _I, _P]),
    "sat_allreduce_bucket": (_I, [_P, _P, _L, _I, _I, _P]),
    "sat_allreduce_finalize": (_I, [_P]),
    # stft.hip
    "sat_fir": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "sat_stft_tiles": (_I, [_I, _I, _I]),
    "sat_stft_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sat_stft_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sat_spec_frames": (_I, [_I, _I, _I]),
    "sat_spec_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sat_spec_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    # attention.hip
    "sat_attn_prepare": (_I, [_P, _L, _L, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sat_attention_fwd": (_I, [_P] * 8 + [_I] * 8 + [_F, _I, _P]),
    "sat_attention_rowdot": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "sat_attention_bwd": (_I, [_P] * 6 + [_I] * 8 + [_F, _I, _P]),
    # edge_conv.hip
    "sat_edge_conv_ok": (_I, [_I] * 6),
    "sat_edge_conv_partial_rows": (_I, [_I] * 2),
    "sat_edge_conv": (_I, [_P] * 15 + [_I] * 9 + [_P]),
    "sat_edge_conv_wgrad_nsplit": (_I, [_I] * 4),
    "sat_edge_conv_wgrad": (_I, [_P] * 6 + [_I] * 6 + [_P]),
    "sat_attention_cross_ok": (_I, [_I] * 5),
    "sat_attention_cross_fwd": (_I, [_P] * 5 + [_I] * 8 + [_F, _P]),
    "sat_attention_cross_bwd_ws": (_L, [_I] * 5),
    "sat_attention_cross_bwd": (_I, [_P] * 7 + [_L] + [_I] * 8 + [_F, _P]),
    # gemm.hip
    "sat_gemm_bf16": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _L, _I, _P, _L, _P] + [_I] * 7 + [_P]),
    "sat_gemm_qkv_bf16": (_I, [_P, _L, _P, _L, _P, _I, _P, _P, _P, _P] + [_I] * 8 + [_P]),
    "sat_gemm_fp8": (_I, [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _L, _I, _P, _L, _P, _P, _P, _P] + [_I] * 6 + [_P]),
    "sat_gemm_qkv_fp8": (_I, [_P, _L, _P, _L, _P, _I, _P, _P, _P, _P, _P, _P, _P] + [_I] * 8 + [_P]),
    "sat_quant_fp8": (_I, [_P, _L, _P, _L, _P, _I, _I, _I, _P]),
    "sat_quant_fp8_rows": (_I, [_P, _L, _P, _L, _P, _I, _I, _I, _P]),
    "sat_absmax_scale": (_I, [_P, _L, _P, _P, _I, _I, _I, _P]),
    "sat_absmax_scale_blocks": (_I, [_I, _I]),
    "sat_splitk_epilogue": (_I, [_P, _I, _P, _P, _L, _P, _L, _I, _I, _I, _P]),
    "sat_cast_bf16": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _I, _P]),
    "sat_cast_bf16_tpair": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _I, _I, _I, _I, _P]),
    "sat_cast_bf16_dual": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _P]),
    "sat_split_bf16x3": (_I, [_P, _L, _P, _L, _I, _I, _I, _P]),
    # dit_ops.hip
    "sat_layernorm_fwd": (_I, [_P] * 5 + [_L] + [_P] * 3 + [_I] * 3 + [_F, _I, _P]),
    "sat_layernorm_fwd_fp8": (_I, [_P] * 5 + [_L] + [_P] * 2 + [_I] * 3 + [_F, _I, _P]),
    "sat_layernorm_bwd_nblocks": (_I, [_I, _I]),
    "sat_layernorm_bwd": (_I, [_P] * 5 + [_L] + [_P] * 4 + [_I] * 4 + [_P]),
    "sat_layernorm_bwd_res": (_I, [_P] * 5 + [_L] + [_P] * 5 + [_I] * 4 + [_P]),
    "sat_rope_tables": (_I, [_P, _P, _I, _I, _F, _P]),
    "sat_rope_apply": (_I, [_P, _P, _L, _L, _L] + [_I] * 7 + [_P]),
    "sat_swiglu": (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
    "sat_gate_residual": (_I, [_P, _P, _L, _P, _P, _I, _I, _I, _I, _P]),
    "sat_gate_residual_bwd_nchunks": (_I, [_I]),
    "sat_cfg_step": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _F, _F, _I, _P]),
    "sat_sampler_step": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _I, _P]),
    "sat_sampler_step_dev": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _I, _P]),
    "sat_gate_residual_bwd": (_I, [_P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _P]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsat_amd.so")


def bind(cdll):
    """Attach restype/argtypes for every C-ABI symbol; raises AttributeError if one is missing."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


_lib = None


def load():
    """Load the gfx950 library (once).  Fails loudly; never substitutes anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"stable_audio_tools_amd: HIP library not built: {LIB_PATH} is missing. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    try:
        cdll = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. no ROCm runtime on this machine
        raise RuntimeError(f"stable_audio_tools_amd: cannot load {LIB_PATH}: {e}. There is no CPU fallback.") from e
    bind(cdll)
    if cdll.sat_is_simulator():
        raise RuntimeError("stable_audio_tools_amd: refusing to use a simulator build as the product library")
    _lib = cdll
    return _lib
