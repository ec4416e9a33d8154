"""Pre-encoded latent datasets: the native VAE encode written in the reference's on-disk format
(pre_encode.py:70-120) and read back with the reference's item semantics (data/dataset.py:265-360)."""
import json
import os

import numpy as np
import pytest
import torch

import seeded
from golden_util import build_native_ae, rel_err


def _case(device, tmp_path):
    from stable_audio_tools_amd.pre_encode import PreEncodedDataset, PreEncoder
    model = build_native_ae("tiny", 100, device)
    ratio = seeded.AE_CONFIGS["tiny"]["model"]["downsampling_ratio"]
    audio = torch.from_numpy(seeded.seeded_array((2, 2, 512), 11, scale=0.5)).to(device)
    pm = torch.ones(2, 512)
    pm[1, 300:] = 0
    md = [{"padding_mask": pm[0], "seconds_total": 3.0, "prompt": "a"}, {"padding_mask": pm[1], "seconds_total": 2.0, "prompt": "b"}]
    enc = PreEncoder(model, tmp_path, rank=0, details={"sample_size": 512})
    torch.manual_seed(5)
    paths = enc.encode_batch(audio, md, batch_idx=7)
    assert [os.path.basename(p) for p in paths] == ["0000000070000.npy", "0000000070001.npy"]       # f"{rank:03d}{batch:06d}{i:04d}"
    assert json.load(open(os.path.join(tmp_path, "details.json"))) == {"sample_size": 512}
    torch.manual_seed(5)
    with torch.no_grad():
        ref = model.encode(audio).cpu()
    lat = np.load(paths[1])
    assert lat.shape == (4, 512 // ratio) and lat.dtype == np.float32 and rel_err(lat, ref[1]) < 1e-6
    info = json.load(open(paths[1][:-4] + ".json"))
    expect = torch.nn.functional.interpolate(pm[1].reshape(1, 1, -1), size=512 // ratio, mode="nearest").squeeze().int().tolist()
    assert info["padding_mask"] == expect and info["prompt"] == "b"
    ds = PreEncodedDataset(str(tmp_path), latent_crop_length=32)
    assert len(ds) == 2
    z, meta = ds[1]
    assert z.shape == (4, 32) and meta["latent_crop_start"] == 0 and meta["padding_mask"][0].shape == (32,)
    assert rel_err(meta["audio"], ref[1][:, :32]) < 1e-6


def test_pre_encode_roundtrip_simulator(emu_modules, tmp_path):
    _case("cpu", tmp_path)


@pytest.mark.gpu
def test_pre_encode_roundtrip_gpu(hip, tmp_path):
    _case("cuda", tmp_path)
