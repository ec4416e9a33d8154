"""A few launches of one DiT projection GEMM shape for rocprofv3 --pmc passes (tools/pmc_gemm.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_audio_tools_amd import ops as O  # noqa: E402

name, m, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n, k = {"qkv": (4608, 1536), "out": (1536, 1536), "ff1": (12288, 1536), "ff2": (1536, 6144)}[name]
ops = O.get_ops()
ops.gemm_tile = tile
a = torch.randn(m, k, device="cuda").bfloat16()
b = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
for _ in range(6):
    c = ops.gemm_bf16(a, b)
torch.cuda.synchronize()
