"""EXPERIMENT: variants of the bf16 attention forward (tools/exp/attn_x.hip -> tools/exp/libattn_x*.so) against the product kernel on the
same operand planes: max error vs fp32 SDPA (random and 'spiky' inputs that force the stale-running-max path) and microseconds per launch.
    python tools/attn_x_bench.py > gpurun_out/attn_x.jsonl"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from stable_audio_tools_amd.ops import get_ops, _ptr  # noqa: E402

o = get_ops()
HERE = os.path.dirname(os.path.abspath(__file__))
LIBS = {}
for tag, name in (("", "libattn_x.so"), ("n", "libattn_x_noslp.so")):
    path = os.path.join(HERE, "exp", name)
    if os.path.exists(path):
        lib = ctypes.CDLL(path)
        lib.satx_attention_fwd.restype = ctypes.c_int
        lib.satx_attention_fwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_void_p]
        LIBS[tag] = lib
VARIANTS = [15, 500, 501]


def timeit(f, n=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run_x(lib, variant, planes, b, h, hkv, nq, nk, out, lse):
    q, k, vt = planes["q"]["rm"][0], planes["k"]["rm"][0], planes["v"]["tr"][0]
    rc = lib.satx_attention_fwd(variant, _ptr(q), _ptr(k), _ptr(vt), _ptr(out), _ptr(lse), b, h, hkv, nq, nk, planes["q"]["np"], planes["k"]["np"],
                                0.125, None)
    assert rc == 0, rc


def case(b, h, hkv, nq, nk, spiky, time_it=True):
    torch.manual_seed(0)
    q = torch.randn(b, h, nq, 64, device="cuda")
    k = torch.randn(b, hkv, nk, 64, device="cuda")
    v = torch.randn(b, hkv, nk, 64, device="cuda")
    if spiky:        # scores that keep growing along the key axis: the running max of the first tiles goes stale again and again
        ramp = torch.linspace(0.2, 6.0, nk, device="cuda").view(1, 1, nk, 1)
        k = k * ramp
        q = q * 1.5
    q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
    out, lse, planes = o.attention(q, k, v, 0.125, return_planes=True)
    rep = h // hkv
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(rep, 1), v.float().repeat_interleave(rep, 1), scale=0.125)
    ref = ref.permute(0, 2, 1, 3).reshape(b, nq, h * 64)
    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), k.float().repeat_interleave(rep, 1)) * 0.125
    lse_ref = torch.logsumexp(s, dim=-1)
    del s

    def err(a):
        return float((a.float() - ref).abs().max() / ref.abs().max())
    rows = [{"kernel": "product", "err": err(out), "lse_err": float((lse - lse_ref).abs().max())}]
    if time_it:
        rows[0]["us"] = round(timeit(lambda: o.attention_planes(planes["q"]["rm"][0], planes["k"]["rm"][0], planes["v"]["tr"][0], nq, nk, 0.125)), 2)
    for tag, lib in LIBS.items():
        for var in VARIANTS:
            ox = torch.zeros_like(out)
            lx = torch.zeros_like(lse)
            run_x(lib, var, planes, b, h, hkv, nq, nk, ox, lx)
            torch.cuda.synchronize()
            r = {"kernel": f"x{var}{tag}", "err": err(ox), "lse_err": float((lx - lse_ref).abs().max())}
            if time_it:
                r["us"] = round(timeit(lambda: run_x(lib, var, planes, b, h, hkv, nq, nk, ox, lx)), 2)
            rows.append(r)
    for r in rows:
        if "us" in r:
            r["tflops"] = round(4.0 * b * h * nq * nk * 64 / r["us"] * 1e-6, 1)
        print(json.dumps({"shape": [b, h, hkv, nq, nk], "spiky": spiky, **r}), flush=True)


for shape in [(2, 24, 24, 1025, 1025), (2, 24, 12, 1025, 130), (2, 24, 24, 6145, 6145), (8, 24, 24, 1025, 1025), (1, 4, 4, 200, 40), (1, 4, 2, 77, 333), (1, 2, 2, 70, 65), (1, 2, 1, 33, 130), (2, 2, 2, 129, 64)]:
    case(*shape, spiky=False, time_it=shape[3] >= 1025)
for shape in [(2, 24, 24, 1025, 1025), (1, 4, 2, 77, 333), (1, 2, 2, 300, 6145)]:
    case(*shape, spiky=True, time_it=False)
