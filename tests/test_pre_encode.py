"""Pre-encoded latent datasets: the native VAE encode written in the reference's on-disk format
(pre_encode.py:70-120), pinned against the REFERENCE'S OWN READER (data/dataset.py:265-360):

  * tests/golden/pre_encoded_reader.npz holds the items the reference's `PreEncodedDataset` returned for a directory written by
    the native `PreEncoder` from seeded inputs (oracle/gen_golden_preencode.py, build container).  Here the native PreEncoder
    writes the same case again (simulator / GPU) and the native reader's items are compared with those reference-read items:
    file names, latents (1e-5: simulator vs GPU arithmetic), every metadata field, with and without a latent crop.
  * where the reference checkout is importable (`-m "not gpu"`, build container) the reference reader is also run LIVE on the
    directory just written and must agree with the native reader exactly.
"""
import json
import os

import numpy as np
import pytest
import torch

import refimport
import seeded
from gen_golden_preencode import CASE, case_inputs
from golden_util import GOLDEN, build_native_ae, rel_err


def _golden():
    z = np.load(os.path.join(GOLDEN, "pre_encoded_reader.npz"))
    doc = json.loads(bytes(z["doc"]).decode())
    assert doc["case"] == CASE, "regenerate tests/golden/pre_encoded_reader.npz (oracle/gen_golden_preencode.py)"
    return z, doc


def _jsonable_item(latents, info):
    info = dict(info)
    assert info.pop("audio") is latents
    name = os.path.basename(info.pop("latent_filename"))
    info["padding_mask"] = [t.tolist() for t in info["padding_mask"]]
    return name, info


def _write(device, tmp_path):
    from stable_audio_tools_amd.pre_encode import PreEncoder
    model = build_native_ae(CASE["ae"], CASE["seed"], device)
    audio, noise, md = case_inputs(device)
    enc = PreEncoder(model, tmp_path, rank=CASE["rank"], details={"sample_size": CASE["length"]})
    paths = enc.encode_batch(audio, md, batch_idx=CASE["batch_idx"], noise=noise)
    return model, audio, noise, paths


def _case(device, tmp_path):
    from stable_audio_tools_amd.pre_encode import PreEncodedDataset
    z, doc = _golden()
    model, audio, noise, paths = _write(device, tmp_path)
    assert [os.path.basename(p) for p in paths] == ["0000000070000.npy", "0000000070001.npy"]       # f"{rank:03d}{batch:06d}{i:04d}"
    assert json.load(open(os.path.join(tmp_path, "details.json"))) == {"sample_size": CASE["length"]}
    with torch.no_grad():
        direct = model.encode(audio, noise=noise).cpu()
    for tag, crop in (("full", None), ("crop", CASE["crop"])):
        ds = PreEncodedDataset(str(tmp_path), latent_crop_length=crop)
        assert len(ds) == CASE["batch"]
        seen = set()
        for i in range(len(ds)):
            latents, info = ds[i]
            name, info = _jsonable_item(latents, info)
            seen.add(name)
            gold_lat, gold_info = z[f"{tag}/{name}"], doc["items"][f"{tag}/{name}"]
            assert latents.dtype == torch.float32 and tuple(latents.shape) == gold_lat.shape
            assert rel_err(latents, gold_lat) < 1e-5, (tag, name)          # what the REFERENCE reader returned for this item
            assert info == gold_info, (tag, name, info, gold_info)
            row = int(name[-8:-4])
            want = direct[row] if crop is None else direct[row][:, :crop]
            assert rel_err(latents, want) < 1e-6                           # and the file is the encode itself
        assert seen == {"0000000070000.npy", "0000000070001.npy"}


def test_pre_encode_matches_reference_reader_golden_simulator(emu_modules, tmp_path):
    _case("cpu", tmp_path)


@pytest.mark.gpu
def test_pre_encode_matches_reference_reader_golden_gpu(hip, tmp_path):
    _case("cuda", tmp_path)


@pytest.mark.skipif(not refimport.available(), reason="needs the reference checkout (build container)")
def test_reference_reader_reads_native_directory_live(emu_modules, tmp_path):
    """The reference's PreEncodedDataset, imported as is, on a directory the native PreEncoder has just written."""
    from gen_golden_preencode import read_with_reference
    from stable_audio_tools_amd.pre_encode import PreEncodedDataset
    _write("cpu", tmp_path)
    for crop in (None, CASE["crop"], 48):
        ref_items = read_with_reference(tmp_path, crop)
        ds = PreEncodedDataset(str(tmp_path), latent_crop_length=crop)
        assert len(ds) == len(ref_items) == CASE["batch"]
        for i in range(len(ds)):
            latents, info = ds[i]
            name, info = _jsonable_item(latents, info)
            ref_lat, ref_info = ref_items[name]
            assert np.array_equal(latents.numpy(), ref_lat)
            assert info == ref_info


def test_reader_filters_and_random_crop(emu_modules, tmp_path):
    """Item semantics the golden case does not reach: random crops stay inside the un-padded part, length filters resample."""
    import random

    from stable_audio_tools_amd.pre_encode import PreEncodedDataset
    _write("cpu", tmp_path)
    random.seed(3)
    ds = PreEncodedDataset(str(tmp_path), latent_crop_length=8, random_crop=True)
    for _ in range(8):
        lat, info = ds[1]                                   # item 1: padding from sample 300 -> latent frame 37
        assert lat.shape == (4, 8) and info["latent_crop_start"] + 8 <= 37 and all(v == 1 for v in info["padding_mask"][0].tolist())
    only_long = PreEncodedDataset(str(tmp_path), min_length_sec=2.5)
    for i in range(2):
        assert only_long[i][1]["seconds_total"] == 3.0
