"""TEST INFRASTRUCTURE ONLY — pins the pre-encoded dataset format against the reference's own READER.

Run in the build container (needs /root/reference):   python oracle/gen_golden_preencode.py
  1. the native PreEncoder (stable_audio_tools_amd/pre_encode.py; kernels on the host-side simulator — this is a fixture
     generator, not a product run) writes a directory from seeded inputs (tests/test_pre_encode.py uses the same ones);
  2. the REFERENCE's `PreEncodedDataset` (stable_audio_tools/data/dataset.py:265-360, imported as is; torchaudio / webdataset
     are stubbed, they are not used by this class) reads that directory, with and without `latent_crop_length`;
  3. the items it returns — latents and the info dict — are committed to tests/golden/pre_encoded_reader.npz.
tests/test_pre_encode.py then compares (a) what the native PreEncoder writes on the GPU and (b) what the native reader returns
with these reference-produced items.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import refimport  # noqa: E402
import seeded  # noqa: E402

CASE = {"ae": "tiny", "seed": 100, "batch": 2, "length": 512, "audio_seed": 11, "noise_seed": 12, "batch_idx": 7, "rank": 0, "crop": 32}


def case_inputs(device="cpu"):
    """(audio, vae noise, metadata list) of the pinned case."""
    c = CASE
    cfg = seeded.AE_CONFIGS[c["ae"]]["model"]
    audio = torch.from_numpy(seeded.seeded_array((c["batch"], 2, c["length"]), c["audio_seed"], scale=0.5)).to(device)
    noise = torch.from_numpy(seeded.seeded_array((c["batch"], cfg["latent_dim"], c["length"] // cfg["downsampling_ratio"]), c["noise_seed"])).to(device)
    pm = torch.ones(c["batch"], c["length"])
    pm[1, 300:] = 0
    md = [{"padding_mask": pm[0], "seconds_total": 3.0, "prompt": "a", "path": "x/a.wav"},
          {"padding_mask": pm[1], "seconds_total": 2.0, "prompt": "b", "path": "x/b.wav"}]
    return audio, noise, md


def reference_reader():
    refimport.install_stubs()
    if "webdataset" not in sys.modules:
        sys.modules["webdataset"] = types.ModuleType("webdataset")       # imported at module scope, unused by PreEncodedDataset
    if refimport.REF_ROOT not in sys.path:
        sys.path.insert(0, refimport.REF_ROOT)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        from stable_audio_tools.data.dataset import LocalDatasetConfig, PreEncodedDataset
    return LocalDatasetConfig, PreEncodedDataset


def read_with_reference(path, crop):
    """{basename: (latents ndarray, info dict made JSON-able)} as the reference's PreEncodedDataset returns them."""
    import contextlib
    LocalDatasetConfig, PreEncodedDataset = reference_reader()
    with contextlib.redirect_stdout(sys.stderr):
        ds = PreEncodedDataset([LocalDatasetConfig(id="golden", path=str(path))], latent_crop_length=crop)
    out = {}
    for i in range(len(ds)):
        latents, info = ds[i]
        info = dict(info)
        assert info.pop("audio") is latents
        name = os.path.basename(info.pop("latent_filename"))
        info["padding_mask"] = [t.tolist() for t in info["padding_mask"]]
        out[name] = (latents.numpy(), info)
    return out


def main():
    from emu_util import emu_ops
    from golden_util import build_native_ae
    from stable_audio_tools_amd import functional
    from stable_audio_tools_amd.pre_encode import PreEncoder
    functional._TEST_OPS = emu_ops()
    model = build_native_ae(CASE["ae"], CASE["seed"], "cpu")
    audio, noise, md = case_inputs()
    arrays, doc = {}, {"case": CASE, "items": {}}
    with tempfile.TemporaryDirectory() as tmp:
        PreEncoder(model, tmp, rank=CASE["rank"], details={"sample_size": CASE["length"]}).encode_batch(audio, md, CASE["batch_idx"], noise=noise)
        for tag, crop in (("full", None), ("crop", CASE["crop"])):
            for name, (lat, info) in read_with_reference(tmp, crop).items():
                arrays[f"{tag}/{name}"] = lat
                doc["items"][f"{tag}/{name}"] = info
    arrays["doc"] = np.frombuffer(json.dumps(doc, sort_keys=True).encode(), dtype=np.uint8)
    out = os.path.join(ROOT, "tests", "golden", "pre_encoded_reader.npz")
    np.savez_compressed(out, **arrays)
    print("wrote", out, sorted(doc["items"]))


if __name__ == "__main__":
    main()
