"""k = 7 conv (ResidualUnit) forward / data-gradient at the Oobleck level shapes: microseconds and fp32-equivalent TFLOP/s,
direct kernel (conv1d_bf16x3_k7.h) vs planes kernel (conv1d_bf16x3_k7p.h: pre-pass + conv).  MI355X only."""
import json, sys
import torch
sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops
o = get_ops()
torch.manual_seed(0)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# the ResidualUnit levels of stable_audio_2_0_vae at 2 097 152 samples (encoder and decoder): (channels, time steps, dilation)
shapes = [(128, 2097152, 1), (128, 2097152, 9), (128, 1048576, 3), (256, 262144, 3), (512, 65536, 9), (1024, 8192, 1)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (c, t, dil) in shapes:
    x = torch.randn(1, c, t, device='cuda'); w = torch.randn(c, c, 7, device='cuda') / (7 * c) ** 0.5
    bias = torch.randn(c, device='cuda'); la = torch.randn(c, device='cuda') * 0.3; lb = torch.randn(c, device='cuda') * 0.3
    x2 = torch.randn(1, c, t, device='cuda')
    wp = o.pack_bf16x3(w, 0, 1)
    pad = 3 * dil
    row = {"C": c, "T": t, "dil": dil}
    flops = 2.0 * c * c * 7 * t
    for name, flag in (("direct", False), ("planes", True)):
        o.k7_planes = flag
        us = timeit(lambda: o.conv1d_bf16x3(x, wp, c, 7, 1, dil, pad, bias=bias, snake=(la, lb)))
        usb = timeit(lambda: o.conv1d_bf16x3(x, wp, c, 7, 1, dil, pad, dsnake=(x2, la, lb)))
        row[name] = {"fwd_us": round(us, 1), "fwd_tf": round(flops / us * 1e-6, 1), "dgrad_us": round(usb, 1), "dgrad_tf": round(flops / usb * 1e-6, 1)}
    print(json.dumps(row), flush=True)
