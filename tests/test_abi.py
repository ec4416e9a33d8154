"""C-ABI contract checks that need no GPU: the gfx950 library loads, exports every symbol that
include/sat_amd.h declares, the ctypes binding table matches the header symbol-for-symbol and
argument-count-for-argument-count, argument validation returns an error code + message instead of
launching, and the product path refuses to run without CUDA(HIP) tensors."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "sat_amd.h")


def _header_decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"(?:const\s+char\s*\*|long\s+long|int)\s+(sat_\w+)\s*\(([^)]*)\)\s*;", src):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        decls[m.group(1)] = n
    return decls


def _ensure_built():
    from stable_audio_tools_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_header_and_binding_agree():
    from stable_audio_tools_amd import _lib
    decls = _header_decls()
    assert len(decls) >= 20
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name][1]) == n, name


def test_gfx950_library_loads_and_exports_every_symbol():
    _lib = _ensure_built()
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_decls():
        assert hasattr(cdll, name), name
    lib = _lib.load()
    assert lib.sat_abi_version() == 1 and lib.sat_is_simulator() == 0


def test_library_contains_gfx950_code_object():
    _lib = _ensure_built()
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for kern in (b"sat_conv1d_kernel", b"sat_convtr1d_kernel", b"sat_conv_wgrad_kernel", b"sat_stft_fwd_kernel",
                 b"sat_stft_bwd_kernel", b"sat_adamw_kernel"):
        assert kern in blob, kern


def test_argument_validation_never_launches():
    lib = _ensure_built().load()
    null = ctypes.c_void_p(None)
    # empty shapes / bad geometry are rejected before any launch, so this is safe without a GPU
    assert lib.sat_conv1d(*([null] * 12), 0, 4, 4, 16, 16, 7, 1, 1, 3, 0, null) != 0
    assert b"empty" in lib.sat_last_error()
    assert lib.sat_conv1d(*([null] * 12), 1, 4, 4, 16, 16, 4, 2, 3, 1, 0, null) != 0
    assert b"dilation" in lib.sat_last_error()
    assert lib.sat_convtr1d(*([null] * 12), 1, 4, 4, 16, 32, 5, 2, 1, 0, null) != 0
    assert b"2*stride" in lib.sat_last_error()
    assert lib.sat_stft_fwd(null, null, null, null, 1, 3, 4096, 1, 1024, 256, null) != 0
    assert lib.sat_stft_fwd(null, null, null, null, 1, 2, 4096, 1, 1000, 250, null) != 0
    assert b"power of two" in lib.sat_last_error()
    assert lib.sat_fir(null, null, null, 1, 100, 100, 0, null) != 0
    assert lib.sat_adamw_step(null, null, null, null, 0, 1e-3, 0.9, 0.99, 1e-8, 0.0, 1, 1.0, null, 0.0, null) != 0
    assert lib.sat_stft_tiles(2048, 512, 2097152) == 1025     # 4097 frames, four per workgroup (eight until round 6's sweep)
    assert lib.sat_stft_tiles(2048, 512, 1000) == -1        # reflect pad needs T > n_fft/2
    assert lib.sat_convtr1d_partial_rows(1, 64, 16, 8) == -1
    # bf16x3 family: geometry rules, plan sizes and partial-plane sizes (host code only)
    assert lib.sat_conv1d_bf16x3(*([null] * 13), 1, 8, 8, 64, 64, 9, 1, 1, 4, 0, null) != 0
    assert b"K <= 8" in lib.sat_last_error()
    assert lib.sat_conv1d_bf16x3(*([null] * 13), 1, 8, 8, 64, 21, 6, 3, 1, 2, 0, null) != 0      # K = 2*stride but stride not 2^n
    assert lib.sat_conv1d_bf16x3(*([null] * 13), 1, 8, 8, 64, 64, 7, 1, 12, 36, 0, null) != 0
    assert b"receptive field" in lib.sat_last_error()
    assert lib.sat_convtr1d_bf16x3(*([null] * 13), 1, 8, 8, 16, 64, 6, 4, 2, 0, null) != 0
    assert b"2*stride" in lib.sat_last_error()
    assert lib.sat_pack_weights_bf16x3_size(128, 128, 7, 1, 0) == 16 * 128 * 64       # 16 chunks of 8 channels x 8 tap groups
    assert lib.sat_pack_weights_bf16x3_size(128, 128, 1, 1, 0) == 4 * 128 * 32        # 4 chunks of 32 channels
    assert lib.sat_pack_weights_bf16x3_size(256, 128, 4, 2, 0) == 8 * 256 * 64        # 128*2 virtual channels, 2 taps
    assert lib.sat_pack_weights_bf16x3_size(128, 128, 6, 3, 0) == -1                  # stride 3: fp32-MFMA fallback kernels
    assert lib.sat_conv1d_bf16x3_partial_rows(1, 2097152, 7, 1) == 8192               # k7: 256-wide time tiles
    assert lib.sat_conv1d_bf16x3_partial_rows(1, 2097152, 1, 1) == 16384              # others: 128-wide
    assert lib.sat_conv_wgrad_bf16x3_nsplit(1, 128, 128, 2097152, 1, 1) == 512
    assert lib.sat_conv_wgrad_bf16x3_nsplit(1, 128, 128, 65536, 3, 1) == -1
    assert lib.sat_conv_wgrad7_bf16x3_fuses_rowsum(1, 128, 2, 2097152) == 1             # narrow input: 4-wave kernel
    assert lib.sat_conv_wgrad7_bf16x3_fuses_rowsum(1, 128, 128, 2097152) == 0           # pipelined kernel: separate sat_rowsum (default)
    assert lib.sat_conv_wgrad7_bf16x3(*([null] * 5), 7, 1, 0, 1, 8, 8, 64, 2, 6, null, null) != 0
    assert b"dilation" in lib.sat_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_has_no_cpu_fallback():
    from stable_audio_tools_amd import functional, ops
    assert not hasattr(functional, "_TEST_OPS"), "the product dispatch must not carry a test hook"
    assert ops.get_ops.__module__ == "stable_audio_tools_amd.ops"
    o = ops.get_ops()
    assert not o.simulator
    with pytest.raises(RuntimeError, match="no CPU path"):
        o.fir(torch.zeros(1, 64), torch.zeros(5))
    from stable_audio_tools_amd.autoencoders import OobleckEncoder
    enc = OobleckEncoder(in_channels=2, channels=8, latent_dim=8, c_mults=[1, 2], strides=[2, 4], use_snake=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(torch.zeros(1, 2, 64))


def test_out_of_scope_configurations_fail_loudly():
    from stable_audio_tools_amd.autoencoders import OobleckDecoder, OobleckEncoder, create_autoencoder_from_config
    with pytest.raises(NotImplementedError):
        OobleckEncoder(use_snake=False)
    with pytest.raises(NotImplementedError):
        OobleckDecoder(use_snake=True, use_nearest_upsample=True)
    with pytest.raises(NotImplementedError):
        create_autoencoder_from_config({"sample_rate": 1, "model": {"encoder": {"type": "dac", "config": {}}, "decoder": {"type": "oobleck", "config": {}},
                                                                     "latent_dim": 1, "downsampling_ratio": 1, "io_channels": 1}})
