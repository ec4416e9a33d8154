"""The 1x1 convs of the ResidualUnits (autoencoders.py:58-83) at the five Oobleck levels of the headline size, on the generic bf16x3 kernel
(plan <4, 4>): forward (SnakeBeta prologue, bias, residual, plane emission for the next unit) and data-gradient (dsnake epilogue).  One JSON
line per (level, kind): microseconds per launch, algorithmic TB/s (three fp32 streams [+ the emitted planes]), fraction of the bf16x3 peak.
    python tools/k1_bench.py [level ...]"""
import json
import sys

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops

o = get_ops()
torch.manual_seed(0)
T0 = 2097152
LEVELS = [(128, T0), (128, T0 // 2), (256, T0 // 8), (512, T0 // 32), (1024, T0 // 256)]


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


for li in ([int(a) for a in sys.argv[1:]] or range(len(LEVELS))):
    c, t = LEVELS[li]
    dev = 'cuda'
    h = torch.randn(1, c, t, device=dev) * 0.5
    x = torch.randn(1, c, t, device=dev) * 0.5
    w = torch.randn(c, c, 1, device=dev) / c ** 0.5
    bias = torch.randn(c, device=dev) * 0.1
    s2 = (torch.randn(c, device=dev) * 0.2, torch.randn(c, device=dev) * 0.2)
    s3 = (torch.randn(c, device=dev) * 0.2, torch.randn(c, device=dev) * 0.2)
    pf, pd = o.pack_bf16x3(w), o.pack_bf16x3(w, mode=1)
    c2 = o.snake_consts(*s2)
    emit = {"snake": s3} if o.emit_ok(c, 1, 1, t, 1) else None
    flops = 2.0 * c * c * t
    kinds = {
        "fwd": (lambda: o.conv1d_bf16x3(h, pf, c, 1, 1, 1, 0, bias=bias, snake=s2, res=x, sconsts=c2, emit=emit), 4.0 * c * t * (4 if emit else 3)),
        "fwd_nosnake": (lambda: o.conv1d_bf16x3(h, pf, c, 1, 1, 1, 0, bias=bias, res=x), 4.0 * c * t * 3),
        "fwd_plain": (lambda: o.conv1d_bf16x3(h, pf, c, 1, 1, 1, 0, bias=bias), 4.0 * c * t * 2),
        "dgrad": (lambda: o.conv1d_bf16x3(x, pd, c, 1, 1, 1, 0, dsnake=(h, *s2)), 4.0 * c * t * 3),
    }
    for name, (fn, nbytes) in kinds.items():
        us = timeit(fn)
        print(json.dumps({"level": li, "c": c, "t": t, "kind": name, "us": round(us, 1), "alg_TBps": round(nbytes / (us * 1e-6) / 1e12, 2),
                          "frac_bf16x3": round(flops / (us * 1e-6) / 833.3e12, 3)}), flush=True)
