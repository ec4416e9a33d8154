"""Self-attention forward (bf16 planes, the sampler's / train step's path): microseconds and achieved TFLOP/s per launch for the two
kernel shapes of sat_attention_fwd (32 / 64 queries per wave, round 6), and what the library picks by itself.
    python tools/attn_bench.py            one JSON line per (shape, kernel)"""
import json
import sys

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops

o = get_ops()
torch.manual_seed(0)


def timeit(f, n=100):
    """microseconds per call of f: 20 calls captured into ONE HIP graph, the graph replayed n / 20 times — a launch of a few microseconds
    is otherwise timed together with the host's ~10 us of Python / ctypes dispatch per call (the first version of this tool did)."""
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                f()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    reps = max(1, n // 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (20 * reps) * 1e3


for (b, h, hkv, nq, nk) in [(2, 24, 24, 1025, 1025), (4, 24, 24, 1025, 1025), (8, 24, 24, 1025, 1025), (16, 24, 24, 1025, 1025), (2, 24, 24, 6145, 6145)]:
    q = torch.randn(b, h, nq, 64, device='cuda').bfloat16()
    k = torch.randn(b, hkv, nk, 64, device='cuda').bfloat16()
    v = torch.randn(b, hkv, nk, 64, device='cuda').bfloat16()
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float(), scale=0.125).permute(0, 2, 1, 3).reshape(b, nq, h * 64)
    for q64 in (False, True, None):
        o.attn_q64 = q64
        out, lse, planes = o.attention(q, k, v, 0.125, return_planes=True)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        us = timeit(lambda: o.attention_planes(planes["q"]["rm"][0], planes["k"]["rm"][0], planes["v"]["tr"][0], nq, nk, 0.125),
                    n=100 if nq < 4000 else 30)
        tf = 4.0 * b * h * nq * nk * 64 / us * 1e-6
        print(json.dumps({"shape": [b, h, hkv, nq, nk], "queries_per_wave": {False: 32, True: 64, None: "auto"}[q64], "us": round(us, 2),
                          "tflops": round(tf, 1), "frac_of_2500": round(tf / 2500, 4), "err": round(err, 5)}), flush=True)
o.attn_q64 = None
