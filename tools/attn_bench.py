"""Self- / cross-attention forward (bf16 planes, the sampler's path): microseconds and achieved TFLOP/s per launch.
SAT_ATTN_FWD_GEN=1 python tools/attn_bench.py  -> first-generation kernel;  default -> attention_fwd64.h."""
import json, os, sys
import torch
sys.path.insert(0, '.')
from stable_audio_tools_amd.ops import get_ops
o = get_ops()
torch.manual_seed(0)
def timeit(f, n=100):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (b, h, hkv, nq, nk) in [(2, 24, 24, 1025, 1025), (2, 24, 12, 1025, 130), (8, 24, 24, 1025, 1025), (2, 24, 24, 6145, 6145)]:
    q = torch.randn(b, h, nq, 64, device='cuda').bfloat16(); k = torch.randn(b, hkv, nk, 64, device='cuda').bfloat16()
    v = torch.randn(b, hkv, nk, 64, device='cuda').bfloat16()
    out, lse, planes = o.attention(q, k, v, 0.125, return_planes=True)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(h // hkv, 1), v.float().repeat_interleave(h // hkv, 1), scale=0.125)
    err = float((out.float() - ref.permute(0, 2, 1, 3).reshape(b, nq, h * 64)).abs().max() / ref.abs().max())
    us = timeit(lambda: o.attention_planes(planes["q"]["rm"][0], planes["k"]["rm"][0], planes["v"]["tr"][0], nq, nk, 0.125))
    print(json.dumps({"gen": os.environ.get("SAT_ATTN_FWD_GEN", "2"), "shape": [b, h, hkv, nq, nk], "us": round(us, 2),
                      "tflops": round(4.0 * b * h * nq * nk * 64 / us * 1e-6, 1), "err": err}))
