#!/usr/bin/env python
"""Times csrc/gemm.hip on the DiT projection shapes (M = model batch x 1025 tokens) against torch.matmul (hipBLASLt) on
the same operands.  Builder-side probe: prints one JSON line per shape / tile / epilogue."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_audio_tools_amd import _lib, ops as O  # noqa: E402

if os.environ.get("SAT_EXP_LIB"):       # experiment builds of the library (tools/exp/): same C-ABI, different compile flags
    _lib.LIB_PATH = os.path.abspath(os.environ["SAT_EXP_LIB"])


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


TILES = tuple(int(t) for t in os.environ.get('SAT_TILES', '0,4').split(','))
SPLITS = tuple(int(t) for t in os.environ.get('SAT_SPLITS', '2,3,4,5,6').split(','))
FP8_TILES = tuple(int(t) for t in os.environ.get('SAT_FP8_TILES', '0,4,7,8').split(','))


def main():
    ops = O.get_ops()
    dev = "cuda"
    torch.manual_seed(0)
    ms = [int(a) for a in sys.argv[1:]] or [2050, 4100]
    shapes = [("qkv", 4608, 1536), ("out", 1536, 1536), ("ff1", 12288, 1536), ("ff2", 1536, 6144), ("kv", 1536, 768)]
    for m in ms:
        for name, n, k in shapes:
            mm = 260 * (m // 2050) if name == "kv" else m
            a = torch.randn(mm, k, device=dev).bfloat16()
            b = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
            res = torch.randn(mm, n, device=dev).bfloat16()
            fl = 2.0 * mm * n * k
            t_ref = timeit(lambda: torch.matmul(a, b.t()))
            row = {"shape": name, "M": mm, "N": n, "K": k, "hipblaslt_us": round(t_ref, 1), "hipblaslt_tf": round(fl / t_ref / 1e6, 1)}
            for tile in TILES:
                ops.gemm_tile = tile
                t = timeit(lambda: ops.gemm_bf16(a, b))
                row[f"native_t{tile}_us"] = round(t, 1)
                row[f"native_t{tile}_tf"] = round(fl / t / 1e6, 1)
                if name in ("out", "ff2"):
                    row[f"native_t{tile}_res_us"] = round(timeit(lambda: ops.gemm_bf16(a, b, res=res, epilogue=ops.EPI_RES)), 1)
                if name == "ff1":
                    row[f"native_t{tile}_swiglu_us"] = round(timeit(lambda: ops.gemm_bf16(a, b, epilogue=ops.EPI_SWIGLU)), 1)
                if name in ("out", "ff2"):
                    for sp in SPLITS:      # split-K slabs + fused bias / residual epilogue kernel
                        row[f"native_t{tile}_splitk{sp}_res_us"] = round(timeit(lambda: ops.gemm_bf16_splitk(a, b, sp, res=res)), 1)
            ops.gemm_tile = None
            # fp8 e4m3 projections (BASELINE.json configs[4]): the MX-MFMA GEMM alone, the activation quantisation pass that feeds it
            # (amax reduction + sat_quant_fp8), and their sum — what a fp8 nn.Linear costs per call (weights are quantised once)
            if k % 16 == 0:
                qa, sa_ = ops.quant_fp8(a)
                qb, sb_ = ops.quant_fp8(b)
                alpha = (sa_ * sb_)
                tq = timeit(lambda: ops.quant_fp8(a))
                row["fp8_quant_act_us"] = round(tq, 1)
                for tile in FP8_TILES:
                    ops.gemm_fp8_tile = tile
                    t8 = timeit(lambda: ops.gemm_fp8(qa, qb, alpha))
                    row[f"fp8_t{tile}_us"] = round(t8, 1)
                    row[f"fp8_t{tile}_tf"] = round(fl / t8 / 1e6, 1)
                    if name == "ff1":
                        row[f"fp8_t{tile}_swiglu_us"] = round(timeit(lambda: ops.gemm_fp8(qa, qb, alpha, epilogue=ops.EPI_SWIGLU)), 1)
                ops.gemm_fp8_tile = None
                row["fp8_pick"] = ops._pick_tile_fp8(mm, n, k)
            row["pick"] = ops._pick_tile(mm, n, 1, k)
            row["pick_splits"] = ops.splitk_for(mm, n, k)
            if name == "qkv" and mm % 1025 == 0:
                # the in-model form of this projection: head split + rotary + attention-plane layout in the epilogue (sat_gemm_qkv_bf16)
                nbat = mm // 1025
                inv = 1.0 / (10000 ** (torch.arange(0, 32, 2, device=dev).float() / 32))
                cs = ops.rope_tables(inv, 1025)
                for tile in TILES:
                    ops.gemm_tile = tile
                    row[f"native_t{tile}_heads_us"] = round(timeit(lambda: ops.gemm_heads_bf16(a, b, cs, 24, nbat, 1025, 0, 3, reuse="bench")), 1)
                ops.gemm_tile = None
            print(json.dumps(row), flush=True)


def train_shapes():
    """The backward GEMMs of a DiT train step at per-GPU batch 4 (M = 4100 tokens): data gradients (bf16 out) and weight gradients
    (fp32 out, reduction over the zero-padded token dim; the bias gradient rides as 8 extra columns) — linear.LinearFn.backward."""
    ops = O.get_ops()
    dev = "cuda"
    torch.manual_seed(0)
    mt = 4104
    shapes = [("dgrad_qkv", 4100, 1536, 4608, False), ("dgrad_out", 4100, 1536, 1536, False), ("dgrad_ff1", 4100, 1536, 12288, False),
              ("dgrad_ff2", 4100, 6144, 1536, False),
              ("wgrad_qkv", 4608, 1544, mt, True), ("wgrad_out", 1536, 1544, mt, True), ("wgrad_ff1", 12288, 1544, mt, True),
              ("wgrad_ff2", 1536, 6152, mt, True)]
    for name, m, n, k, f32 in shapes:
        a = torch.randn(m, k, device=dev).bfloat16()
        b = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
        fl = 2.0 * m * n * k
        t_ref = timeit(lambda: torch.matmul(a, b.t()))
        row = {"shape": name, "M": m, "N": n, "K": k, "f32_out": f32, "hipblaslt_us": round(t_ref, 1)}
        for tile in TILES:
            ops.gemm_tile = tile
            row[f"native_t{tile}_us"] = round(timeit(lambda: ops.gemm_bf16(a, b, out_dtype=torch.float32 if f32 else torch.bfloat16)), 1)
            if f32:
                for sp in (2, 4):
                    row[f"native_t{tile}_splits{sp}_us"] = round(timeit(lambda: ops.gemm_bf16(a, b, out_dtype=torch.float32, splits=sp)), 1)
        ops.gemm_tile = None
        row["pick"] = ops._pick_tile(m, n, 1, k)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    if os.environ.get("SAT_BENCH_TRAIN") == "1":
        train_shapes()
    else:
        main()
