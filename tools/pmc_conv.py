"""Run ONE conv configuration a few times (for rocprofv3 --pmc runs).  usage: pmc_conv.py <kind> [C] [T]
kind: conv7 | conv7ns | dgrad7 | conv1 | wgrad7 | wgrad1 | calib"""
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
elif kind == "calib":
    # known-traffic calibration launches: sat_rowsum reads C*T*4 bytes with 16-byte loads; the torch copy reads and
    # writes C*T*4 bytes
    fn = lambda: (ops.rowsum(x), x.clone())
else:
    raise SystemExit("unknown kind")
for _ in range(4):
    fn()
torch.cuda.synchronize()
