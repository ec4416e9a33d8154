"""Pretransform adaptor in front of the diffusion model — the drop-in boundary named in
BASELINE.json (reference: stable_audio_tools/models/pretransforms.py:6-27 Pretransform ABC,
:29-89 AutoencoderPretransform)."""
from torch import nn


class Pretransform(nn.Module):
    def __init__(self, enable_grad, io_channels, is_discrete):
        super().__init__()
        self.is_discrete = is_discrete
        self.io_channels = io_channels
        self.encoded_channels = None
        self.downsampling_ratio = None
        self.enable_grad = enable_grad

    def encode(self, x):
        raise NotImplementedError

    def decode(self, z):
        raise NotImplementedError

    def tokenize(self, x):
        raise NotImplementedError

    def decode_tokens(self, tokens):
        raise NotImplementedError


class AutoencoderPretransform(Pretransform):
    """Frozen autoencoder; encode divides by `scale`, decode multiplies (pretransforms.py:51-74).
    `model_half` (fp16 weights) is not offered: the HIP conv stack computes in fp32 on the f32 matrix
    cores, which is the reference's default numerics for the VAE."""

    def __init__(self, model, scale=1.0, model_half=False, iterate_batch=False, chunked=False):
        super().__init__(enable_grad=False, io_channels=model.io_channels,
                         is_discrete=model.bottleneck is not None and model.bottleneck.is_discrete)
        if model_half:
            raise NotImplementedError("model_half is not supported on the HIP path (fp32 conv stack)")
        self.model = model
        self.model.requires_grad_(False).eval()
        self.scale = scale
        self.downsampling_ratio = model.downsampling_ratio
        self.io_channels = model.io_channels
        self.sample_rate = model.sample_rate
        self.model_half = False
        self.iterate_batch = iterate_batch
        self.encoded_channels = model.latent_dim
        self.chunked = chunked
        self.num_quantizers = None
        self.codebook_size = None

    def encode(self, x, **kwargs):
        encoded = self.model.encode_audio(x, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)
        return encoded / self.scale

    def decode(self, z, **kwargs):
        z = z * self.scale
        return self.model.decode_audio(z, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)

    def tokenize(self, x, **kwargs):
        raise AssertionError("Cannot tokenize with a continuous model")

    def decode_tokens(self, tokens, **kwargs):
        raise AssertionError("Cannot decode tokens with a continuous model")

    def load_state_dict(self, state_dict, strict=True):
        return self.model.load_state_dict(state_dict, strict=strict)
