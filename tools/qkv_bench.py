"""Self-attention input projection (M = nb x ntok tokens, 1536 -> 3 x 24 x 64): plain-store GEMM vs the fused head-split / rotary /
plane-layout epilogue (sat_gemm_qkv_bf16), per batch layout."""
import json, os, sys
import torch
sys.path.insert(0, '.')
from stable_audio_tools_amd import _lib
from stable_audio_tools_amd.ops import get_ops
if os.environ.get('SAT_EXP_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SAT_EXP_LIB'])
o = get_ops()
torch.manual_seed(0)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (nb, ntok) in [(2, 1025), (2, 1024), (4, 1025), (2, 6145)]:
    m = nb * ntok
    x = torch.randn(m, 1536, device='cuda').bfloat16()
    w = (torch.randn(4608, 1536, device='cuda') / 39).bfloat16()
    inv = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    cs = o.rope_tables(inv.cuda(), ntok)
    row = {"nb": nb, "ntok": ntok,
           "plain_us": round(timeit(lambda: o.gemm_bf16(x, w)), 1),
           "qkv_rope_us": round(timeit(lambda: o.gemm_heads_bf16(x, w, cs, 24, nb, ntok, 0, 3, reuse="self")), 1),
           "q_only_us": round(timeit(lambda: o.gemm_heads_bf16(x, w[:1536], None, 24, nb, ntok, 0, 1, reuse="cross")), 1),
           "q_plain_us": round(timeit(lambda: o.gemm_bf16(x, w[:1536])), 1)}
    print(json.dumps(row), flush=True)
