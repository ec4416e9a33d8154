"""Offline pre-encoding of audio into VAE latents with the native Oobleck encoder, in the reference's on-disk format
 — the producer side of `pre_encoded: true` DiT training (training/diffusion.py:376-379).

Reference: pre_encode.py:40-124 (`PreEncodedLatentsInferenceWrapper.validation_step`: one `<rank>/<id>.npy` latent (C, N) +
`<id>.json` metadata per item, id = f"{rank:03d}{batch_idx:06d}{i:04d}", padding mask nearest-resampled to the latent
length, `details.json` at the root) and data/dataset.py:265-360 (`PreEncodedDataset`: the consumer).  The Lightning /
dataloader / CLI shell around it is out of scope (host-side I/O); what is native here is the encode itself.
"""
import json
import os
import random

import numpy as np
import torch
from torch.nn import functional as F


def _jsonable(v):
    if isinstance(v, torch.Tensor):
        return v.cpu().numpy().tolist()
    if isinstance(v, np.ndarray):
        return v.tolist()
    return v


class PreEncoder:
    """Encodes batches of audio with `model.encode` (an AudioAutoencoder or anything with the same method) and writes them
    as the reference's pre-encoded dataset.  `model_half` (pre_encode.py:31-32 `model.to(torch.float16)`, :84-85 the audio cast): the
    model's parameters are stored in fp16 and the audio is rounded to fp16 as in the reference; the HIP conv stack computes at fp32
    accuracy on those rounded values (autoencoders._WNConvBase.p32) and the latents are written as fp16, the dtype the reference's
    half model produces."""

    def __init__(self, model, output_path, rank=0, is_discrete=False, model_half=False, details=None):
        if is_discrete:
            raise NotImplementedError("discrete (token) bottlenecks are out of scope")
        self.model_half = bool(model_half)
        if self.model_half:
            model.half()
        self.model = model
        self.output_path = str(output_path)
        self.rank = int(rank)
        os.makedirs(os.path.join(self.output_path, str(self.rank)), exist_ok=True)
        dpath = os.path.join(self.output_path, "details.json")
        if self.rank == 0 and not os.path.exists(dpath):            # pre_encode.py:57-68: written once, by rank 0
            with open(dpath, "w") as f:
                json.dump(details if details is not None else {}, f)

    @torch.no_grad()
    def encode_batch(self, audio, metadata, batch_idx, **encode_kwargs):
        """audio (B, C, T) on the model's device; metadata: list of B dicts with at least `padding_mask` (T,) (tensor / list).
        encode_kwargs go to model.encode (e.g. noise= for a reproducible VAE draw).  Returns the list of written latent paths."""
        if audio.ndim == 4 and audio.shape[0] == 1:                 # pre_encode.py:77-78
            audio = audio[0]
        if self.model_half:
            audio = audio.half().float()
            latents = self.model.encode(audio, **encode_kwargs).half().cpu().numpy()
        else:
            latents = self.model.encode(audio, **encode_kwargs).float().cpu().numpy()
        paths = []
        for i, latent in enumerate(latents):
            latent_id = f"{self.rank:03d}{batch_idx:06d}{i:04d}"
            base = os.path.join(self.output_path, str(self.rank), latent_id)
            with open(base + ".npy", "wb") as f:
                np.save(f, latent)
            md = dict(metadata[i])
            pm = torch.as_tensor(md["padding_mask"]).float().reshape(1, 1, -1)
            md["padding_mask"] = F.interpolate(pm, size=latent.shape[1], mode="nearest").squeeze().int().cpu().numpy().tolist()
            md = {k: _jsonable(v) for k, v in md.items()}
            with open(base + ".json", "w") as f:
                json.dump(md, f)
            paths.append(base + ".npy")
        return paths


def latent_filenames(path, extension="npy"):
    out = []
    for root, _, files in os.walk(path):
        out.extend(os.path.join(root, f) for f in files if f.endswith("." + extension) and not f.startswith("."))
    return sorted(out)


class PreEncodedDataset(torch.utils.data.Dataset):
    """Reader of the format above with the reference's item semantics (data/dataset.py:265-360): returns (latents (C, N),
    info) with info["padding_mask"] = [tensor], optional crop of `latent_crop_length` frames (random start inside the
    un-padded part when `random_crop`), length filters on info["seconds_total"]."""

    def __init__(self, paths, latent_crop_length=None, min_length_sec=None, max_length_sec=None, random_crop=False, latent_extension="npy"):
        super().__init__()
        self.latent_extension = latent_extension
        self.filenames = []
        for p in ([paths] if isinstance(paths, (str, os.PathLike)) else paths):
            self.filenames.extend(latent_filenames(str(p), latent_extension))
        self.latent_crop_length, self.random_crop = latent_crop_length, random_crop
        self.min_length_sec, self.max_length_sec = min_length_sec, max_length_sec

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        fn = self.filenames[idx]
        latents = torch.from_numpy(np.load(fn))
        with open(fn[:-len(self.latent_extension) - 1] + ".json") as f:
            info = json.load(f)
        info["latent_filename"] = fn
        if self.latent_crop_length is not None:
            pm = info["padding_mask"]
            last_ix = len(pm) - 1 - pm[::-1].index(1)
            start = random.randint(0, last_ix - self.latent_crop_length) if (self.random_crop and last_ix > self.latent_crop_length) else 0
            latents = latents[:, start:start + self.latent_crop_length]
            info["padding_mask"] = pm[start:start + self.latent_crop_length]
            info["latent_crop_length"], info["latent_crop_start"] = self.latent_crop_length, start
        info["padding_mask"] = [torch.tensor(info["padding_mask"])]
        sec = info.get("seconds_total")
        if sec is not None and ((self.min_length_sec is not None and sec < self.min_length_sec)
                                or (self.max_length_sec is not None and sec > self.max_length_sec)):
            return self[random.randrange(len(self))]
        info["audio"] = latents
        return latents, info
