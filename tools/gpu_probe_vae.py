"""GPU probe: per-kernel timing of the Oobleck conv stack at BASELINE.json configs[1] shapes
(47.55 s stereo 44.1 kHz -> T = 2097152).  Prints TFLOP/s per layer type and end-to-end encode /
decode / train-step-without-loss timing.  Scratch tool (results go to gpurun_out/)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_audio_tools_amd import functional as Fn  # noqa: E402
from stable_audio_tools_amd import ops as O  # noqa: E402

ops = O.get_ops()
dev = "cuda"


def timeit(fn, iters=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


res = {}
T0 = int(os.environ.get("SAT_T", 2097152))
levels = [(128, T0, 2), (128, T0 // 2, 4), (256, T0 // 8, 4), (512, T0 // 32, 8), (1024, T0 // 256, 8)]
for (C, T, S) in levels:
    x = torch.randn(1, C, T, device=dev) * 0.5
    la = torch.randn(C, device=dev) * 0.1
    lb = torch.randn(C, device=dev) * 0.1
    w7 = torch.randn(C, C, 7, device=dev) / (C * 7) ** 0.5
    w1 = torch.randn(C, C, 1, device=dev) / C ** 0.5
    bias = torch.randn(C, device=dev) * 0.1
    wp7 = ops.pack(w7, O.PACK_CONV_FWD)
    wp1 = ops.pack(w1, O.PACK_CONV_FWD)
    for dil in (1, 9):
        ms = timeit(lambda: ops.conv1d(x, wp7, C, 7, 1, dil, 3 * dil, bias=bias, snake=(la, lb)))
        fl = 2 * C * C * 7 * T
        res[f"conv7_d{dil}_C{C}_T{T}"] = dict(ms=ms, tflops=fl / ms / 1e9)
    planes = ops.pack_bf16x3(w7)
    for dil in (1, 9):
        ms = timeit(lambda: ops.conv1d_bf16x3(x, planes, C, 7, 1, dil, 3 * dil, bias=bias, snake=(la, lb)))
        res[f"conv7x3_d{dil}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * 7 * T / ms / 1e9)
    y_a = ops.conv1d(x, wp7, C, 7, 1, 9, 27, bias=bias, snake=(la, lb))
    y_b = ops.conv1d_bf16x3(x, planes, C, 7, 1, 9, 27, bias=bias, snake=(la, lb))
    res[f"conv7x3_vs_f32_relerr_C{C}_T{T}"] = dict(err=float((y_a - y_b).abs().max() / y_a.abs().max()))
    del y_a, y_b
    ms = timeit(lambda: ops.conv1d(x, wp1, C, 1, 1, 1, 0, bias=bias, snake=(la, lb), res=x))
    res[f"conv1_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * T / ms / 1e9, gbps=3 * 4 * C * T / ms / 1e6)
    p1 = ops.pack_bf16x3(w1)
    ms = timeit(lambda: ops.conv1d_bf16x3(x, p1, C, 1, 1, 1, 0, bias=bias, snake=(la, lb), res=x))
    res[f"conv1x3_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * T / ms / 1e9, gbps=3 * 4 * C * T / ms / 1e6)
    # down conv C -> 2C
    wd = torch.randn(2 * C, C, 2 * S, device=dev) / (C * 2 * S) ** 0.5
    wpd = ops.pack(wd, O.PACK_CONV_FWD)
    ms = timeit(lambda: ops.conv1d(x, wpd, 2 * C, 2 * S, S, 1, (S + 1) // 2, snake=(la, lb)))
    res[f"down_s{S}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * 2 * C * 2 * S * (T // S) / ms / 1e9)
    pd = ops.pack_bf16x3(wd, stride=S)
    ms = timeit(lambda: ops.conv1d_bf16x3(x, pd, 2 * C, 2 * S, S, 1, (S + 1) // 2, snake=(la, lb)))
    res[f"downx3_s{S}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * 2 * C * 2 * S * (T // S) / ms / 1e9)
    # up conv 2C -> C  (input at T/S)
    xu = torch.randn(1, 2 * C, T // S, device=dev) * 0.5
    la2 = torch.randn(2 * C, device=dev) * 0.1
    wu = torch.randn(2 * C, C, 2 * S, device=dev) / (2 * C * 2) ** 0.5
    wpu = ops.pack(wu, O.PACK_POLYPHASE, S)
    ms = timeit(lambda: ops.convtr1d(xu, wpu, C, 2 * S, S, (S + 1) // 2, snake=(la2, la2)))
    res[f"up_s{S}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * 2 * C * 2 * S * (T // S) / ms / 1e9)
    pu = ops.pack_bf16x3(wu, mode=2, stride=S)
    ms = timeit(lambda: ops.convtr1d_bf16x3(xu, pu, C, 2 * S, S, (S + 1) // 2, snake=(la2, la2)))
    res[f"upx3_s{S}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * 2 * C * 2 * S * (T // S) / ms / 1e9)
    # backward pieces for the k7 conv
    dy = torch.randn(1, C, T, device=dev)
    ms = timeit(lambda: ops.conv_wgrad(dy, x, 7, 1, 9, 27, snake=(la, lb), snake_on=2))
    res[f"wgrad7_d9_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * 7 * T / ms / 1e9)
    for dil in (1, 9):
        ms = timeit(lambda: ops.conv_wgrad7_bf16x3(dy, x, dil, 3 * dil, snake=(la, lb)))
        res[f"wgrad7x3_d{dil}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * 7 * T / ms / 1e9)
    ms = timeit(lambda: ops.conv_wgrad(dy, x, 1, 1, 1, 0, snake=(la, lb), snake_on=2))
    res[f"wgrad1_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * T / ms / 1e9)
    dyd = torch.randn(1, 2 * C, T // S, device=dev)
    ms = timeit(lambda: ops.conv_wgrad(dyd, x, 2 * S, S, 1, (S + 1) // 2, snake=(la, lb), snake_on=2))
    res[f"wgrad_down_s{S}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * 2 * C * 2 * S * (T // S) / ms / 1e9)
    xu2 = torch.randn(1, 2 * C, T // S, device=dev) * 0.5
    ms = timeit(lambda: ops.conv_wgrad(xu2, dy, 2 * S, S, 1, (S + 1) // 2, snake=(la2, la2), snake_on=1))
    res[f"wgrad_up_s{S}_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * 2 * C * 2 * S * (T // S) / ms / 1e9)
    del dyd, xu2
    wpb = ops.pack(w7, O.PACK_CONV_DGRAD)
    ms = timeit(lambda: ops.conv1d(dy, wpb, C, 7, 1, 9, 27, dsnake=(x, la, lb), res=dy))
    res[f"dgrad7_d9_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * 7 * T / ms / 1e9)
    planes_b = ops.pack_bf16x3(w7, mode=1)
    ms = timeit(lambda: ops.conv1d_bf16x3(dy, planes_b, C, 7, 1, 9, 27, dsnake=(x, la, lb), res=dy))
    res[f"dgrad7x3_d9_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * 7 * T / ms / 1e9)
    p1b = ops.pack_bf16x3(w1, mode=1)
    ms = timeit(lambda: ops.conv1d_bf16x3(dy, p1b, C, 1, 1, 1, 0, dsnake=(x, la, lb)))
    res[f"dgrad1x3_C{C}_T{T}"] = dict(ms=ms, tflops=2 * C * C * T / ms / 1e9, gbps=3 * 4 * C * T / ms / 1e6)
    ms = timeit(lambda: ops.rowsum(dy))
    res[f"rowsum_C{C}_T{T}"] = dict(ms=ms, gbps=4 * C * T / ms / 1e6)
    del x, dy, xu
    torch.cuda.empty_cache()
    for k, v in res.items():
        if f"_C{C}_T{T}" in k:
            print(k, {a: round(b, 3) for a, b in v.items()}, flush=True)

# edge convs
x2 = torch.randn(1, 2, T0, device=dev)
w = torch.randn(128, 2, 7, device=dev)
wp = ops.pack(w, O.PACK_CONV_FWD)
ms = timeit(lambda: ops.conv1d(x2, wp, 128, 7, 1, 1, 3))
res["first_conv_2_128"] = dict(ms=ms, gbps=4 * 130 * T0 / ms / 1e6)
xl = torch.randn(1, 128, T0, device=dev)
w = torch.randn(2, 128, 7, device=dev)
wp = ops.pack(w, O.PACK_CONV_FWD)
la = torch.zeros(128, device=dev)
ms = timeit(lambda: ops.conv1d(xl, wp, 2, 7, 1, 1, 3, snake=(la, la)))
res["last_conv_128_2"] = dict(ms=ms, gbps=4 * 130 * T0 / ms / 1e6)
pl = ops.pack_bf16x3(w)
ms = timeit(lambda: ops.conv1d_bf16x3(xl, pl, 2, 7, 1, 1, 3, snake=(la, la)))
res["last_conv_128_2_x3"] = dict(ms=ms, gbps=4 * 130 * T0 / ms / 1e6)
w = torch.randn(128, 2, 7, device=dev)
pf = ops.pack_bf16x3(w)
ms = timeit(lambda: ops.conv1d_bf16x3(x2, pf, 128, 7, 1, 1, 3))
res["first_conv_2_128_x3"] = dict(ms=ms, gbps=4 * 130 * T0 / ms / 1e6)
print("x3 first", res["first_conv_2_128_x3"], "last", res["last_conv_128_2_x3"], flush=True)
print("first", res["first_conv_2_128"], "last", res["last_conv_128_2"], flush=True)
del x2, xl
torch.cuda.empty_cache()

# end-to-end through the product modules
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config  # noqa: E402

cfg = json.load(open(os.path.join(ROOT, "stable_audio_tools_amd", "configs", "stable_audio_2_0_vae.json")))
model = create_autoencoder_from_config(cfg).to(dev)
audio = torch.randn(1, 2, T0, device=dev) * 0.1
with torch.no_grad():
    t = timeit(lambda: model.encode(audio), iters=2)
    res["encode_fwd_ms"] = t
    z = model.encode(audio)
    t = timeit(lambda: model.decode(z), iters=2)
    res["decode_fwd_ms"] = t
print("encode fwd ms", res["encode_fwd_ms"], "decode fwd ms", res["decode_fwd_ms"], flush=True)
torch.cuda.reset_peak_memory_stats()


def step():
    z, info = model.encode(audio, return_info=True)
    dec = model.decode(z)
    loss = dec.square().mean() + 1e-4 * info["kl"]
    loss.backward()


t = timeit(step, iters=2)
res["fwd_bwd_noloss_ms"] = t
res["peak_mem_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
print("fwd+bwd ms", t, "peak GB", res["peak_mem_gb"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe_vae.json"), "w"), indent=1)
