"""The strided (down) and transposed (up) convs of the Oobleck stack at the headline size (autoencoders.py:233-283), each with its
data-gradient — the 20 launches per generator step the generic bf16x3 kernel (csrc/conv1d_bf16x3.hip, plans <8, 4>) serves.
One JSON line per (level, kind): microseconds per launch, fraction of the bf16x3 matrix peak (833 TFLOP/s) and algorithmic TB/s
(input + output [+ x2 of the data-gradient], fp32).  MI355X only.
    python tools/strided_bench.py [level ...]"""
import json
import sys

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops

o = get_ops()
torch.manual_seed(0)
T0 = 2097152
# (Cin, Cout, stride, Tin) of the encoder's down convs; the decoder's up convs are their mirror images
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


want = [int(a) for a in sys.argv[1:]] or list(range(len(LEVELS)))
for li in want:
    ci, co, s, tin = LEVELS[li]
    k, pad, tout = 2 * s, (s + 1) // 2, tin // s
    dev = 'cuda'
    x = torch.randn(1, ci, tin, device=dev) * 0.5          # the wide-time tensor (down conv's input / up conv's output side)
    z = torch.randn(1, co, tout, device=dev) * 0.5         # the short-time tensor
    w_dn = torch.randn(co, ci, k, device=dev) / (ci * k) ** 0.5
    w_up = torch.randn(co, ci, k, device=dev) / (co * 2) ** 0.5          # ConvTranspose1d weight [in = co][out = ci][K]
    b_dn = torch.randn(co, device=dev) * 0.1
    b_up = torch.randn(ci, device=dev) * 0.1
    sx = (torch.randn(ci, device=dev) * 0.2, torch.randn(ci, device=dev) * 0.2)
    sz = (torch.randn(co, device=dev) * 0.2, torch.randn(co, device=dev) * 0.2)
    p_dn = o.pack_bf16x3(w_dn, stride=s)
    p_dn_t = o.pack_bf16x3(w_dn, mode=2, stride=s)
    p_up = o.pack_bf16x3(w_up, mode=2, stride=s)
    p_up_d = o.pack_bf16x3(w_up, stride=s)
    cx, cz = o.snake_consts(*sx), o.snake_consts(*sz)
    emit = {"snake": sz} if o.emit_ok(co, k, s, tout, 1) else None
    flops = 2.0 * co * ci * k * tout
    kinds = {
        "down_fwd": (lambda: o.conv1d_bf16x3(x, p_dn, co, k, s, 1, pad, bias=b_dn, snake=sx, sconsts=cx, emit=emit), 4.0 * (ci * tin + co * tout)),
        "down_dgrad": (lambda: o.convtr1d_bf16x3(z, p_dn_t, ci, k, s, pad, tout=tin, dsnake=(x, *sx)), 4.0 * (2 * ci * tin + co * tout)),
        "up_fwd": (lambda: o.convtr1d_bf16x3(z, p_up, ci, k, s, pad, bias=b_up, snake=sz, sconsts=cz), 4.0 * (ci * tin + co * tout)),
        "up_dgrad": (lambda: o.conv1d_bf16x3(x, p_up_d, co, k, s, 1, pad, dsnake=(z, *sz), tout=tout), 4.0 * (ci * tin + 2 * co * tout)),
    }
    for name, (fn, nbytes) in kinds.items():
        us = timeit(fn)
        print(json.dumps({"level": li, "cin": ci, "cout": co, "stride": s, "tin": tin, "kind": name, "us": round(us, 1),
                          "frac_bf16x3": round(flops / (us * 1e-6) / 833.3e12, 3), "alg_TBps": round(nbytes / (us * 1e-6) / 1e12, 2)}), flush=True)
