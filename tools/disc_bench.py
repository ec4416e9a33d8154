#!/usr/bin/env python
"""Times csrc/disc_conv.hip on the MS-STFT discriminator's layer shapes for one 47.55 s stereo item (B = 1, T = 2097152): per scale
(n_fft, hop) the 64 -> 64 (3 x 9) conv forward, its data-gradient, its weight-gradient and the planes / dpre pass, against round 2's
virtual-channel path (conv2d_virtual: row packing + the 1-D conv kernels).  Builder-side probe: one JSON line per scale."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_audio_tools_amd import _lib, ops as O  # noqa: E402

if os.environ.get("SAT_EXP_LIB"):       # experiment builds of the library (tools/exp/): same C-ABI
    _lib.LIB_PATH = os.path.abspath(os.environ["SAT_EXP_LIB"])
from stable_audio_tools_amd.discriminators import conv2d_virtual  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps      # ms


def main():
    ops = O.get_ops()
    torch.manual_seed(0)
    T = int(os.environ.get("SAT_DISC_T", 2097152))
    scales = [(2048, 512), (1024, 256), (512, 128), (256, 64), (128, 32)]
    if len(sys.argv) > 1:
        scales = [s for s in scales if str(s[0]) in sys.argv[1:]]
    c = 64
    for n_fft, hop in scales:
        frames, wd = (T - n_fft) // hop + 1, n_fft // 2 + 1
        P, L, lead, rows = ops.disc_geom(frames, wd)
        for kh, kw, dil in ((3, 9, 2), (3, 3, 1)):
            x = torch.randn(1, c, L, device="cuda") * 0.5
            w4 = torch.randn(c, c, kh, kw, device="cuda") * 0.05
            bias = torch.randn(c, device="cuda")
            _, xp = ops.disc_planes(x, frames, wd, slot=0)
            wq, wqt = ops.disc_pack(w4, 0), ops.disc_pack(w4, 1)
            fl = 2.0 * c * c * kh * kw * frames * wd
            row = {"n_fft": n_fft, "frames": frames, "freq": wd, "L": L, "kernel": [kh, kw], "gflop": round(fl / 1e9, 1)}
            t = timeit(lambda: ops.disc_conv(xp, wq, bias, 1, c, c, frames, wd, kh, kw, dil, 0.2, emit_slot=1))
            row["conv_emit_ms"], row["conv_emit_tf"] = round(t, 3), round(fl / t / 1e9, 1)
            t = timeit(lambda: ops.disc_conv(xp, wqt, None, 1, c, c, frames, wd, kh, kw, dil, 1.0))
            row["dgrad_ms"], row["dgrad_tf"] = round(t, 3), round(fl / t / 1e9, 1)
            y = torch.randn(1, c, L, device="cuda")
            t = timeit(lambda: ops.disc_wgrad(y, x, frames, wd, kh, kw, dil))
            row["wgrad_ms"], row["wgrad_tf"] = round(t, 3), round(fl / t / 1e9, 1)
            t = timeit(lambda: ops.disc_planes(y, frames, wd, out=x, slope=0.2, want_dst=True, slot=0))
            row["dpre_pass_ms"] = round(t, 3)
            t = timeit(lambda: ops.rowsum(y))
            row["rowsum_ms"] = round(t, 3)
            del y, xp
            if os.environ.get("SAT_DISC_OLD", "1") == "1" and n_fft >= 512:
                x4 = torch.randn(1, c, frames, wd, device="cuda") * 0.5
                with torch.no_grad():
                    t = timeit(lambda: conv2d_virtual(x4, w4, bias, dil_t=dil, pad_t=dil * (kh - 1) // 2, slope=0.2), reps=3, warm=1)
                row["round2_fwd_ms"] = round(t, 3)
                del x4
            print(json.dumps(row), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
