"""Upper bound of what batching the per-conv weight preparation can buy (round 6 experiment, numerics deliberately WRONG in the second
arm): the generator step of bench.py timed (a) as shipped and (b) with ops.wn_fold / pack_bf16x3 / pack_k7q / pack / snake_consts
memoised on (storage pointer, arguments) — i.e. the ~300 tiny launches per step that only depend on the parameters are gone, every other
kernel runs as before on the stale derived weights.  A / B / A / B, 6 steps each after 2 warm-up.
    python profiles/r06_experiments/prep_ablation/run.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from stable_audio_tools_amd import ops as O
from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
from stable_audio_tools_amd.training import AutoencoderTrainStep

dev = torch.device("cuda", 0)
cfg = json.load(open(os.path.join(ROOT, "stable_audio_tools_amd", "configs", "stable_audio_2_0_vae.json")))
torch.manual_seed(1234)
model = create_autoencoder_from_config(cfg).to(dev)
with torch.no_grad():
    for n_, p in model.named_parameters():
        if n_.endswith("alpha") or n_.endswith("beta"):
            p.normal_(0.0, 0.1)
stepper = AutoencoderTrainStep(model, cfg, use_discriminator=False)
stepper.use_disc = False
ops = O.get_ops()
g = torch.Generator().manual_seed(0)
batches = [(0.1 * torch.randn(1, 2, 2097152, generator=g)).to(dev) for _ in range(2)]
names = ("wn_fold", "pack_bf16x3", "pack_k7q", "pack", "snake_consts")
orig = {n: getattr(ops, n) for n in names}


def memo(fn):
    cache = {}

    def wrapped(*a, **k):
        key = tuple((x.data_ptr(), tuple(x.shape)) if torch.is_tensor(x) else x for x in a) + \
            tuple(sorted((kk, (v.data_ptr() if torch.is_tensor(v) else v)) for kk, v in k.items()))
        if key not in cache:
            cache[key] = fn(*a, **k)
        return cache[key]
    return wrapped


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        stepper(batches[i % 2])
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for i in range(2):
    stepper(batches[i % 2])
res = []
for rep in range(2):
    for arm in ("shipped", "memoised_prep"):
        for n in names:
            setattr(ops, n, memo(orig[n]) if arm == "memoised_prep" else orig[n])
        stepper(batches[0])      # fill the memo / settle
        res.append((arm, round(timed(6), 3)))
print(json.dumps({"ms_per_step": res}))
