"""Which Python lines cause the device-to-device copies / fills / adds of a DiT train step?  One profiled step (torch.profiler, CUDA + CPU
activities, stacks), the small torch-side kernels grouped by (kernel family, aten op, first package frame, input shapes).
    python tools/diag_dit_copies.py [batch]"""
import json
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, '.')
from stable_audio_tools_amd.dit import DiffusionTransformer
from stable_audio_tools_amd.training import DiTTrainStep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
cfg = json.load(open(os.path.join(ROOT, "stable_audio_tools_amd", "configs", "stable_audio_open_dit.json")))
dcfg = dict(cfg["diffusion"]["config"])
dcfg["depth"] = 4
torch.manual_seed(1234)
model = DiffusionTransformer(**dcfg).to(dev).train(True)
stepper = DiTTrainStep(model, lr=5e-5, cfg_dropout_prob=0.1, autocast_dtype=torch.bfloat16)
tlat, m = cfg["latent_length"], cfg["context_length"]
lat = torch.randn(b, dcfg["io_channels"], tlat, device=dev)
cross = torch.randn(b, m, dcfg["cond_token_dim"], device=dev)
glob = torch.randn(b, dcfg["global_cond_dim"], device=dev)
for _ in range(2):
    stepper(lat, cross_attn_cond=cross, global_embed=glob)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    stepper(lat, cross_attn_cond=cross, global_embed=glob)
    torch.cuda.synchronize()
ev = prof.events()
c = Counter()
for e in ev:
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    for k in e.kernels:
        kn = k.name
        fam = ("memcpy" if ("Memcpy" in kn or "copyBuffer" in kn) else "fill" if "FillFunctor" in kn or "fillBuffer" in kn else
               "add" if "CUDAFunctor_add" in kn else "copy_kernel" if "direct_copy" in kn or "copy_kernel" in kn else None)
        if fam is None:
            continue
        st = [s for s in (e.stack or []) if "stable_audio_tools_amd" in s]
        loc = st[0].split("stable_audio_tools_amd/")[-1][:60] if st else "autograd engine / other"
        c[(fam, e.name, loc, str(e.input_shapes)[:70])] += 1
for (fam, name, loc, shp), n in sorted(c.items(), key=lambda kv: -kv[1])[:50]:
    print(n, fam, name, loc, shp)
print("depth", dcfg["depth"], "batch", b)
