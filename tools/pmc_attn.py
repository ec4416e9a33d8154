"""A few launches of the bf16 attention forward for rocprofv3 --pmc passes (tools/pmc_attn.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_audio_tools_amd import ops as O  # noqa: E402

b, n = int(sys.argv[1]), int(sys.argv[2])
ops = O.get_ops()
q = torch.randn(b, 24, n, 64, device="cuda").bfloat16()
k = torch.randn(b, 24, n, 64, device="cuda").bfloat16()
v = torch.randn(b, 24, n, 64, device="cuda").bfloat16()
out, lse, planes = ops.attention(q, k, v, 0.125, return_planes=True)
for _ in range(6):
    ops.attention_planes(planes["q"]["rm"][0], planes["k"]["rm"][0], planes["v"]["tr"][0], n, n, 0.125)
torch.cuda.synchronize()
