"""Run ONE conv configuration a few times (for rocprofv3 --pmc runs).  usage: pmc_conv.py <kind> [C] [T]
kind: conv7 | conv7q | conv7ns | dgrad7 | conv1 | wgrad7 | wgrad1 | wgrads2 | wgrads8 | ruk1 | disc9 | calib"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_audio_tools_amd import ops as O  # noqa: E402

kind = sys.argv[1]
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
T = int(sys.argv[3]) if len(sys.argv) > 3 else 2097152
ops = O.get_ops()
dev = "cuda"
x = torch.randn(1, C, T, device=dev) * 0.5
dy = torch.randn(1, C, T, device=dev)
la = torch.randn(C, device=dev) * 0.1
lb = torch.randn(C, device=dev) * 0.1
bias = torch.randn(C, device=dev) * 0.1
w7 = torch.randn(C, C, 7, device=dev) / (C * 7) ** 0.5
w1 = torch.randn(C, C, 1, device=dev) / C ** 0.5
if kind == "conv7":
    pl = ops.pack_bf16x3(w7)
    fn = lambda: ops.conv1d_bf16x3(x, pl, C, 7, 1, 9, 27, bias=bias, snake=(la, lb))
elif kind == "conv7q":      # conv1d_bf16x3_k7q.h (the shipped kernel: planes pre-pass + q-packed weights)
    ops.k7q = True
    ops.k7q_min_cin = ops.k7q_min_cout = 1
    pl = ops.pack_bf16x3(w7, 0, 1, q=True)
    fn = lambda: ops.conv1d_bf16x3(x, pl, C, 7, 1, 9, 27, bias=bias, snake=(la, lb))
elif kind == "disc9":       # csrc/disc_conv.hip: the discriminator's 64 -> 64 (3 x 9) layer at the n_fft = 1024 scale of a 2 097 152-sample item
    frames, wd = 8189, 513
    L = ops.disc_geom(frames, wd)[1]
    xd = torch.randn(1, 64, L, device=dev) * 0.5
    w4 = torch.randn(64, 64, 3, 9, device=dev) * 0.05
    _, xp = ops.disc_planes(xd, frames, wd, slot=0)
    wq = ops.disc_pack(w4, 0)
    b64 = torch.randn(64, device=dev)
    fn = lambda: ops.disc_conv(xp, wq, b64, 1, 64, 64, frames, wd, 3, 9, 2, 0.2, emit_slot=1)
elif kind == "conv7ns":
    pl = ops.pack_bf16x3(w7)
    fn = lambda: ops.conv1d_bf16x3(x, pl, C, 7, 1, 9, 27, bias=bias)
elif kind == "dgrad7":
    pl = ops.pack_bf16x3(w7, mode=1)
    fn = lambda: ops.conv1d_bf16x3(dy, pl, C, 7, 1, 9, 27, dsnake=(x, la, lb), res=dy)
elif kind == "conv1":
    pl = ops.pack_bf16x3(w1)
    fn = lambda: ops.conv1d_bf16x3(x, pl, C, 1, 1, 1, 0, bias=bias, snake=(la, lb), res=x)
elif kind == "wgrad7":
    fn = lambda: ops.conv_wgrad7_bf16x3(dy, x, 9, 27, snake=(la, lb))
elif kind == "wgrad1":
    fn = lambda: ops.conv_wgrad(dy, x, 1, 1, 1, 0, snake=(la, lb), snake_on=2)
elif kind in ("wgrads2", "wgrads8"):
    # the strided convs' weight gradient (sat_wgrad_small_bf16x3_kernel<2>) at two Oobleck levels: block 0 (128 -> 128, stride 2,
    # output length 1 048 576) and block 3 (512 -> 1024, stride 8, output length 8192)
    if kind == "wgrads2":
        co, ci, tl, s_ = 128, 128, 1048576, 2
    else:
        co, ci, tl, s_ = 1024, 512, 8192, 8
    dyl = torch.randn(1, co, tl, device=dev)
    xh = torch.randn(1, ci, tl * s_, device=dev) * 0.5
    la2 = torch.randn(ci, device=dev) * 0.1
    lb2 = torch.randn(ci, device=dev) * 0.1
    fn = lambda: ops.conv_wgrad(dyl, xh, 2 * s_, s_, 1, (s_ + 1) // 2, snake=(la2, lb2), snake_on=2, lo_rowsum=True)
elif kind == "ruk1":         # csrc/ru_k1_bwd.hip: the whole backward of a unit's 1 x 1 conv (C = 128) in one pass over dy and h
    wt = ops.ru_k1_pack(w1)
    fn = lambda: ops.ru_k1_bwd(dy, x, w1, (la, lb), emit=True, wt=wt, raw=True)
elif kind == "calib":
    # known-traffic calibration launches: sat_rowsum reads C*T*4 bytes with 16-byte loads; the torch copy reads and
    # writes C*T*4 bytes
    fn = lambda: (ops.rowsum(x), x.clone())
else:
    raise SystemExit("unknown kind")
for _ in range(4):
    fn()
torch.cuda.synchronize()
