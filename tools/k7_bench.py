"""k = 7 conv (ResidualUnit) forward / data-gradient at the Oobleck level shapes: microseconds and fp32-equivalent TFLOP/s,
direct kernel (conv1d_bf16x3_k7.h) vs the planes kernel every C >= 64 level takes (conv1d_bf16x3_k7q.h: pre-pass + conv; in the model the
producer's epilogue writes the planes instead of the pre-pass).  MI355X only."""
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
    wq = o.pack_bf16x3(w, 0, 1, q=True)
    for name, flag, wts in (("direct", False, wp), ("q", True, wq)):
        o.k7q = flag
        o.k7q_min_cin = o.k7q_min_cout = 1
        us = timeit(lambda: o.conv1d_bf16x3(x, wts, c, 7, 1, dil, pad, bias=bias, snake=(la, lb)))
        usb = timeit(lambda: o.conv1d_bf16x3(x, wts, c, 7, 1, dil, pad, dsnake=(x2, la, lb)))
        row[name] = {"fwd_us": round(us, 1), "fwd_tf": round(flops / us * 1e-6, 1), "dgrad_us": round(usb, 1), "dgrad_tf": round(flops / usb * 1e-6, 1)}
    # the planes pre-pass alone (SnakeBeta + split of the input), included in the "q" row above
    import ctypes
    rows = o.lib.sat_conv1d_k7_plane_rows(t, t, pad)
    hi = torch.empty(((c + 7) // 8) * rows * 8, dtype=torch.int16, device='cuda'); lo = torch.empty_like(hi)
    sa, sib = o.snake_consts(la, lb)
    P = lambda z: ctypes.c_void_p(z.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    row["prepass_us"] = round(timeit(lambda: o.lib.sat_conv1d_k7_planes(P(x), P(sa), P(sib), P(hi), P(lo), 1, c, t, rows, st)), 1)
    print(json.dumps(row), flush=True)
