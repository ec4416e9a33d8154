"""TEST INFRASTRUCTURE ONLY — golden vectors of the REFERENCE's own samplers (stable_audio_tools/inference/sampling.py) around the
REFERENCE's DiffusionTransformer, run on CPU in the build container:

    sample (v-DDIM: eta 0, eta 0.4, cfg_pp)        :254-307     on "tiny_adaln" (v objective)
    sample_discrete_euler (+ DistributionShift)     :98-135      on "small_rf"   (rectified flow)
    sample_rk4                                      :138-177
    sample_flow_dpmpp                               :179-219
    sample_flow_pingpong                            :222-250
    sample_rf (logSNR schedule + dispatch)          :395-446     euler / dpmpp
    sample_k  ("v-ddim" branch)                     :334-391

Weights, inputs and the noise streams are regenerated from seeds (oracle/seeded.py, oracle/gen_golden.dit_inputs, torch.manual_seed);
only outputs are committed: tests/golden/samplers.npz.  Each entry also records the per-step `denoised` the reference hands to
its callback (first batch item, first channel) so that the fused update's second output is pinned too.

    python oracle/gen_golden_samplers.py
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refimport  # noqa: E402
import seeded  # noqa: E402
from gen_golden import dit_inputs  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

# what each case runs: (model config name, weight seed, sampler name, kwargs).  tests/test_samplers.py reads this table too.
GUIDE_V = dict(cfg_scale=6.0, scale_phi=0.75)
GUIDE_RF = dict(cfg_scale=3.0, scale_phi=0.5)
NOISE_SEED = 4242
CASES = {
    "ddim":         ("tiny_adaln", 710, "sample", dict(steps=5, eta=0.0), GUIDE_V),
    "ddim_eta":     ("tiny_adaln", 710, "sample", dict(steps=4, eta=0.4), GUIDE_V),
    "ddim_cfgpp":   ("tiny_adaln", 710, "sample", dict(steps=4, eta=0.0, cfg_pp=True), GUIDE_V),
    "ddim_shift":   ("tiny_adaln", 710, "sample", dict(steps=4, eta=0.0, sigma_max=0.9, dist_shift="shift"), GUIDE_V),
    "sample_k":     ("tiny_adaln", 710, "sample_k", dict(steps=4, sampler_type="v-ddim", sigma_max=100, device="cpu"), GUIDE_V),
    "euler":        ("small_rf", 720, "sample_discrete_euler", dict(steps=4), GUIDE_RF),
    "euler_shift":  ("small_rf", 720, "sample_discrete_euler", dict(steps=4, dist_shift="shift", sigma_max=0.95), GUIDE_RF),
    "euler_sigmas": ("small_rf", 720, "sample_discrete_euler", dict(sigmas=[0.9, 0.6, 0.35, 0.1, 0.0]), GUIDE_RF),
    "rk4":          ("small_rf", 720, "sample_rk4", dict(steps=2), GUIDE_RF),
    "dpmpp":        ("small_rf", 720, "sample_flow_dpmpp", dict(steps=5), GUIDE_RF),
    "dpmpp_sigmas": ("small_rf", 720, "sample_flow_dpmpp", dict(sigmas=[0.97, 0.8, 0.55, 0.3, 0.12, 0.0]), GUIDE_RF),
    "pingpong":     ("small_rf", 720, "sample_flow_pingpong", dict(steps=3), GUIDE_RF),
    "rf_euler":     ("small_rf", 720, "sample_rf", dict(steps=4, sampler_type="euler", device="cpu"), GUIDE_RF),
    "rf_dpmpp":     ("small_rf", 720, "sample_rf", dict(steps=5, sampler_type="dpmpp", sigma_max=0.8, device="cpu", with_init=True), GUIDE_RF),
    "rf_rk4":       ("small_rf", 720, "sample_rf", dict(steps=2, sampler_type="rk4", device="cpu"), GUIDE_RF),
}


def case_kwargs(kw, shift_cls):
    """Materialise the table's placeholders: "shift" -> a DistributionShift of the calling side's class (small max_length so that the
    37-frame test sequence actually moves the schedule), `sigmas` lists -> fp32 tensors."""
    kw = dict(kw)
    if kw.get("dist_shift") == "shift":
        kw["dist_shift"] = shift_cls(base_shift=0.5, max_shift=1.15, max_length=64, min_length=16)
    if "sigmas" in kw:
        kw["sigmas"] = torch.tensor(kw["sigmas"], dtype=torch.float32)
    kw.pop("with_init", None)
    return kw


def case_inputs(name, with_init=False):
    inp = dit_inputs(name)
    extra = dict(cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"])
    if "prepend_cond" in inp:
        extra.update(prepend_cond=inp["prepend_cond"], prepend_cond_mask=inp["prepend_cond_mask"])
    init = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 611, scale=0.5)) if with_init else None
    return inp["x"], extra, init


def build_reference_model(name, seed):
    from stable_audio_tools.models.dit import DiffusionTransformer
    model = DiffusionTransformer(**seeded.DIT_CONFIGS[name]).float().train(False)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seed).items()}, strict=False)
    return model


def run_case(sampling, model, case):
    """Runs one table entry with `sampling` = a module holding the sampler functions (the reference's or the native one)."""
    name, _, fn_name, kw, guide = CASES[case]
    x, extra, init = case_inputs(name, with_init=kw.get("with_init", False))
    kw = case_kwargs(kw, sampling.DistributionShift)
    seen = []

    def callback(d):
        seen.append(d["denoised"][0, 0].detach().float().cpu().clone())
    fn = getattr(sampling, fn_name)
    torch.manual_seed(NOISE_SEED)                     # the stream torch.randn_like draws from (eta > 0, ping-pong)
    if fn_name in ("sample_rf", "sample_k"):
        out = fn(model, x, init, callback=callback, **kw, **extra, **guide)
    elif fn_name == "sample":
        steps, eta = kw.pop("steps"), kw.pop("eta")
        out = fn(model, x, steps, eta, callback=callback, **kw, **extra, **guide)
    else:
        out = fn(model, x, callback=callback, **kw, **extra, **guide)
    return out, torch.stack(seen)


def main():
    os.makedirs(OUT, exist_ok=True)
    with contextlib.redirect_stdout(sys.stderr):
        refimport.import_reference()
    import stable_audio_tools.inference.sampling as ref_sampling
    torch.set_num_threads(8)
    models, out = {}, {}
    for case, (name, seed, *_rest) in CASES.items():
        if name not in models:
            models[name] = build_reference_model(name, seed)
        with torch.no_grad():
            y, den = run_case(ref_sampling, models[name], case)
        out[case] = y.numpy()
        out[case + "/denoised"] = den.numpy()
        print(f"{case}: out {tuple(y.shape)} |out| {float(y.abs().max()):.4f} callbacks {den.shape[0]}", file=sys.stderr)
    np.savez_compressed(os.path.join(OUT, "samplers.npz"), **out)


if __name__ == "__main__":
    main()
