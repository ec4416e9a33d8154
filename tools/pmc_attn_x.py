"""A few launches of ONE experimental attention-forward variant (tools/exp) for rocprofv3 --pmc passes (tools/pmc_attn_x.sh).
    python tools/pmc_attn_x.py <lib tag: '' | n> <variant> <B> <N>"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_audio_tools_amd.ops import get_ops, _ptr  # noqa: E402

tag, var, b, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
o = get_ops()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libattn_x_noslp.so" if tag == "n" else "libattn_x.so"))
lib.satx_attention_fwd.restype = ctypes.c_int
lib.satx_attention_fwd.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_void_p]
q = torch.randn(b, 24, n, 64, device="cuda").bfloat16()
k = torch.randn(b, 24, n, 64, device="cuda").bfloat16()
v = torch.randn(b, 24, n, 64, device="cuda").bfloat16()
out, lse, planes = o.attention(q, k, v, 0.125, return_planes=True)
ox = torch.empty_like(out)
for _ in range(6):
    rc = lib.satx_attention_fwd(var, _ptr(planes["q"]["rm"][0]), _ptr(planes["k"]["rm"][0]), _ptr(planes["v"]["tr"][0]), _ptr(ox), None, b, 24, 24, n, n,
                                planes["q"]["np"], planes["k"]["np"], 0.125, None)
    assert rc == 0
torch.cuda.synchronize()
