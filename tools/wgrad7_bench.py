"""The k = 7 convs' weight gradient (csrc/conv_wgrad7_bf16x3_pipe.h) at four Oobleck level shapes: microseconds per launch with and without the
SnakeBeta recompute of x, fraction of the bf16x3 peak.  MI355X only.
    python tools/wgrad7_bench.py"""
import json
import sys

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops

o = get_ops()


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


for (c, t, dil) in [(128, 2097152, 1), (128, 2097152, 9), (256, 262144, 3), (512, 65536, 9)]:
    x = torch.randn(1, c, t, device='cuda') * 0.5
    dy = torch.randn(1, c, t, device='cuda')
    la = torch.randn(c, device='cuda') * 0.1
    lb = torch.randn(c, device='cuda') * 0.1
    us = timeit(lambda: o.conv_wgrad7_bf16x3(dy, x, dil, 3 * dil, snake=(la, lb), dy_rowsum=True, raw=True))
    usn = timeit(lambda: o.conv_wgrad7_bf16x3(dy, x, dil, 3 * dil, raw=True))
    print(json.dumps({"C": c, "T": t, "dil": dil, "us": round(us, 1), "nosnake_us": round(usn, 1),
                      "frac": round(2.0 * c * c * 7 * t / us * 1e-6 / 833.3, 3)}), flush=True)
