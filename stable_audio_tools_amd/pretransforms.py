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
    `model_half` (pretransforms.py:39, :48-71): as in the reference the autoencoder's parameters are converted to fp16 (state_dict
    dtype, memory footprint and weight rounding are the reference's), inputs are rounded to fp16 and results come back as fp32
    holding fp16-representable values.  BETWEEN those roundings the HIP conv stack keeps its fp32-accurate arithmetic (bf16x3 split
    products, fp32 accumulation) instead of accumulating in half precision: the output is within fp16 resolution of the reference's
    half path and closer to the fp32 model than that path is (tests/test_boundary.py::test_pretransform_model_half_*)."""

    def __init__(self, model, scale=1.0, model_half=False, iterate_batch=False, chunked=False):
        super().__init__(enable_grad=False, io_channels=model.io_channels,
                         is_discrete=model.bottleneck is not None and model.bottleneck.is_discrete)
        self.model = model
        self.model.requires_grad_(False).eval()
        self.scale = scale
        self.downsampling_ratio = model.downsampling_ratio
        self.io_channels = model.io_channels
        self.sample_rate = model.sample_rate
        self.model_half = bool(model_half)
        self.iterate_batch = iterate_batch
        self.encoded_channels = model.latent_dim
        self.chunked = chunked
        self.num_quantizers = None
        self.codebook_size = None
        if self.model_half:
            self.model.half()

    def encode(self, x, **kwargs):
        if self.model_half:
            x = x.half().float()          # the rounding of the reference's x.half(); the kernels take fp32
        encoded = self.model.encode_audio(x, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)
        if self.model_half:
            encoded = encoded.half().float()
        return encoded / self.scale

    def decode(self, z, **kwargs):
        z = z * self.scale
        if self.model_half:
            z = z.half().float()
        decoded = self.model.decode_audio(z, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)
        if self.model_half:
            decoded = decoded.half().float()
        return decoded

    def tokenize(self, x, **kwargs):
        raise AssertionError("Cannot tokenize with a continuous model")

    def decode_tokens(self, tokens, **kwargs):
        raise AssertionError("Cannot decode tokens with a continuous model")

    def load_state_dict(self, state_dict, strict=True):
        return self.model.load_state_dict(state_dict, strict=strict)
