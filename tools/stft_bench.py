"""The MR-STFT loss kernels (csrc/stft.hip; auraloss.py:368-449) at the headline size: one stereo item of 2 097 152 samples, the four views
(sum / difference / left / right) and the seven resolutions of the autoencoder's spectral loss.  One JSON line per (resolution, kind):
microseconds per launch, algorithmic TB/s (x and y read once per resolution: 2 signals x 2 channels x 4 T bytes; the backward also writes
its four gradient planes once: + 4 x 2 x 4 T bytes) — the figures SURVEY.md section 8(d) prices this path with.
    python tools/stft_bench.py [--lib variant.so] [n_fft ...]
--lib: time a variant build of csrc/stft.hip (tools/stft_sweep.sh) instead of the product library; its forward sums and backward planes are
first compared with the product library's on the same inputs."""
import json
import os
import sys

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops

o = get_ops()
lib = o.lib
argv = sys.argv[1:]
if argv[:1] == ["--lib"]:
    import ctypes
    from stable_audio_tools_amd import _lib as L
    lib = L.bind(ctypes.CDLL(os.path.abspath(argv[1])))
    argv = argv[2:]
torch.manual_seed(0)
T = 2097152
RES = [(2048, 512), (1024, 256), (512, 128), (256, 64), (128, 32), (64, 16), (32, 8)]


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


x = torch.randn(1, 2, T, device='cuda') * 0.1
y = x + 0.01 * torch.randn_like(x)
views = torch.tensor([[1.0, 1.0], [1.0, -1.0], [1.0, 0.0], [0.0, 1.0]], device='cuda')
coef = torch.rand(1, 4, 3, device='cuda') * 1e-3
planes = torch.zeros(4, 1, 2, T, device='cuda')
want = [int(a) for a in argv]
tot = {"fwd": 0.0, "bwd": 0.0}
for n, hop in RES:
    if want and n not in want:
        continue
    tiles = lib.sat_stft_tiles(n, hop, T)
    partial = torch.empty(4 * 3, tiles, device='cuda')
    st = o._stream(x)
    from stable_audio_tools_amd.ops import _ptr
    if lib is not o.lib:      # the variant against the product library
        lib.sat_stft_fwd(_ptr(x), _ptr(y), _ptr(views), _ptr(partial), 1, 2, T, 4, n, hop, st)
        ref = o.stft_sums(x, y, views, n, hop).view(-1)
        got = partial.sum(1)
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5, ("variant sums differ", n)
        pv = torch.zeros_like(planes)
        lib.sat_stft_bwd(_ptr(x), _ptr(y), _ptr(views), _ptr(coef), _ptr(pv), 1, 2, T, 4, n, hop, 0, st)
        pr = torch.zeros_like(planes)
        o.stft_backward(x, y, views, coef, pr, n, hop)
        gv, gr = pv.sum(0), pr.sum(0)
        assert float((gv - gr).abs().max() / gr.abs().max()) < 1e-4, ("variant gradient differs", n)
    kinds = {
        "fwd": (lambda: lib.sat_stft_fwd(_ptr(x), _ptr(y), _ptr(views), _ptr(partial), 1, 2, T, 4, n, hop, st), 4.0 * T * 4),
        "bwd": (lambda: lib.sat_stft_bwd(_ptr(x), _ptr(y), _ptr(views), _ptr(coef), _ptr(planes), 1, 2, T, 4, n, hop, 0, st), 4.0 * T * (4 + 8)),
    }
    for name, (fn, nbytes) in kinds.items():
        us = timeit(fn)
        tot[name] += us
        print(json.dumps({"n_fft": n, "hop": hop, "kind": name, "us": round(us, 1), "alg_TBps": round(nbytes / (us * 1e-6) / 1e12, 3)}), flush=True)
print(json.dumps({"total_us": {k: round(v, 1) for k, v in tot.items()}, "sum_ms": round(sum(tot.values()) / 1e3, 3)}))
