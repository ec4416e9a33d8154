"""Which projections carry the fp8 distance?  Depth-24 SA-Open DiT at N = 6145 (bench.py's long_context weights and inputs), final output
(conditioned half) of the bf16 model with fp8 e4m3 switched on for subsets of the projections, relative L2 against the native fp32 path
(bf16x3 products: 3e-6 from the reference's fp32 model, tests/test_full_width.py) on the same 16-bit-rounded weights.
    python profiles/r06_experiments/fp8_policy/run.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from stable_audio_tools_amd import linear
from stable_audio_tools_amd.dit import DiffusionTransformer
from stable_audio_tools_amd.linear import Linear

dev = torch.device("cuda", 0)
cfg = json.load(open(os.path.join(ROOT, "stable_audio_tools_amd", "configs", "stable_audio_open_dit.json")))
dcfg = cfg["diffusion"]["config"]
torch.manual_seed(1234)
model = DiffusionTransformer(**dcfg)
with torch.no_grad():
    for n_, p_ in model.named_parameters():
        if n_.endswith("to_out.weight") or ".ff.ff.2." in n_ or "process_conv" in n_:
            p_.normal_(0.0, 0.02)
model = model.to(device=dev, dtype=torch.bfloat16).train(False)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, dcfg["io_channels"], 6144, generator=g).to(dev, torch.bfloat16)
cross = torch.randn(1, cfg["context_length"], dcfg["cond_token_dim"], generator=g).to(dev, torch.bfloat16)
glob = torch.randn(1, dcfg["global_cond_dim"], generator=g).to(dev, torch.bfloat16)
t = torch.full((1,), 0.5, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    m32 = DiffusionTransformer(**dcfg).to(dev).train(False)
    m32.load_state_dict({k: v.float() for k, v in model.state_dict().items()}, strict=False)
    ref = m32(x.float(), t.float(), cross_attn_cond=cross.float(), global_embed=glob.float()).float()
    del m32
    torch.cuda.empty_cache()


def run(select):
    n = 0
    for name, m in model.named_modules():
        if isinstance(m, Linear) and min(m.in_features, m.out_features) >= 256 and m.in_features % 16 == 0:
            m.fp8 = bool(select(name))
            n += int(m.fp8)
    with torch.no_grad():
        out = model(x, t, cross_attn_cond=cross, global_embed=glob).float()
    return n, float((out - ref).norm() / ref.norm())


sets = {"none (bf16)": lambda n: False, "all": lambda n: True, "ff pair": lambda n: ".ff." in n, "ff.0.proj (FF1) only": lambda n: ".ff.ff.0." in n,
        "ff.2 (FF2) only": lambda n: ".ff.ff.2" in n, "attention projections": lambda n: ".ff." not in n,
        "self-attn to_qkv only": lambda n: n.endswith("self_attn.to_qkv"), "to_out (self + cross) only": lambda n: n.endswith(".to_out"),
        "cross to_q / to_kv only": lambda n: n.endswith("cross_attn.to_q") or n.endswith("cross_attn.to_kv")}
for label, sel in sets.items():
    n, e = run(sel)
    print(json.dumps({"fp8_on": label, "linears": n, "rel_l2_vs_fp32": round(e, 4)}), flush=True)
