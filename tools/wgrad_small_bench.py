"""The strided convs' weight gradient (sat_wgrad_small_bf16x3_kernel<2>, csrc/conv_wgrad_bf16x3.hip) at the five Oobleck levels of the headline
size: microseconds per launch and fraction of the bf16x3 peak.  MI355X only.
    python tools/wgrad_small_bench.py"""
import json
import sys

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops

o = get_ops()
T0 = 2097152
LEVELS = [(128, 128, 2, T0), (128, 256, 4, T0 // 2), (256, 512, 4, T0 // 8), (512, 1024, 8, T0 // 32), (1024, 2048, 8, T0 // 256)]


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for li, (ci, co, s, tin) in enumerate(LEVELS):
    tl = tin // s
    dy = torch.randn(1, co, tl, device='cuda')
    x = torch.randn(1, ci, tin, device='cuda') * 0.5
    la = torch.randn(ci, device='cuda') * 0.1
    lb = torch.randn(ci, device='cuda') * 0.1
    us = timeit(lambda: o.conv_wgrad(dy, x, 2 * s, s, 1, (s + 1) // 2, snake=(la, lb), snake_on=2, lo_rowsum=True, raw=True))
    print(json.dumps({"level": li, "cin": ci, "cout": co, "stride": s, "tin": tin, "us": round(us, 1),
                      "frac_bf16x3": round(2.0 * co * ci * 2 * s * tl / us * 1e-6 / 833.3, 3)}), flush=True)
