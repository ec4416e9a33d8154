"""MI355X-native Oobleck audio autoencoder — drop-in for the reference's hot-path classes.

Mirrors (same class names, constructor kwargs, module tree and therefore state_dict keys):
  stable_audio_tools/models/autoencoders.py
    WNConv1d :23, WNConvTranspose1d :26, ResidualUnit :58, EncoderBlock :233, DecoderBlock :252,
    OobleckEncoder :285, OobleckDecoder :320, AudioAutoencoder :401 (encode :446, decode :493)
  stable_audio_tools/models/blocks.py  SnakeBeta :301
so a reference checkpoint loads unchanged (weight_g / weight_v / bias / alpha / beta).

What is different is the execution: the module tree is only a parameter container.  Forward walks
it in *fusion units* (functional.py) so SnakeBeta, bias, residual add and tanh never make their own
trip through HBM, and both forward and backward run on the HIP kernels of csrc/.  There is no
PyTorch conv fallback: without the gfx950 library these modules raise.
"""
import copy
import math

import torch
from torch import nn
from torch.nn.utils.weight_norm import WeightNorm as _TorchWeightNorm

from . import functional as Fn
from .bottleneck import VAEBottleneck  # noqa: F401  (re-export, mirrors reference import surface)


class SnakeBeta(nn.Module):
    """Parameter holder for SnakeBeta (blocks.py:301-329): log-scale alpha/beta, zero-init.
    Standalone forward exists for API parity; inside the Oobleck stack it is always fused into the
    following convolution's prologue."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=True):
        super().__init__()
        if not alpha_logscale:
            raise NotImplementedError("only alpha_logscale=True (the reference default used by Oobleck) is implemented")
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        self.alpha = nn.Parameter(torch.zeros(in_features) * alpha)
        self.beta = nn.Parameter(torch.zeros(in_features) * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.beta.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001

    def forward(self, x):
        # identity 1x1 "conv" with the snake prologue: keeps the standalone call on the HIP path
        c = x.shape[1]
        eye = torch.eye(c, device=x.device, dtype=x.dtype).unsqueeze(-1)
        a, b = (self.alpha, self.beta) if self.alpha.dtype == torch.float32 else (self.alpha.detach().float(), self.beta.detach().float())
        return Fn.SnakeConv1dFn.apply(x, a, b, eye, None, None, 1, 1, 0, False)


class _NativeWeightNorm(_TorchWeightNorm):
    """The forward-pre-hook object torch.nn.utils.weight_norm leaves on a module, re-implemented for the native convs so that the
    reference's `remove_weight_norm_from_model` (models/utils.py:31-37; train.py:73-81 `--remove-pretransform-weight-norm`) keeps
    working unchanged: torch.nn.utils.remove_weight_norm looks for a WeightNorm hook named "weight" and calls its .remove().
    Nothing happens per forward (the fold is a HIP launch inside the conv's fusion unit, or cached); .remove() turns the module into
    its folded form: a plain `weight` parameter, no weight_g / weight_v — the state_dict layout of a checkpoint saved after removal."""

    def __init__(self, name="weight", dim=0):
        super().__init__(name, dim)

    def compute_weight(self, module):
        return module._fold_host()

    def remove(self, module):
        # a module that already became folded by LOADING a weight-norm-removed checkpoint still carries this hook object: a later
        # remove_weight_norm / remove_weight_norm_from_model (train.py `--remove-pretransform-weight-norm post_load`) then has nothing
        # to fold — torch deletes the hook right after this call
        if not module.is_folded:
            module._to_folded(module._fold_host().detach())

    def __call__(self, module, inputs):
        return None


def _inference_pass(*tensors):
    """True when no gradient can be asked of this call: derived weights may come from the layer's DerivedCache."""
    return (not torch.is_grad_enabled()) or not any(t is not None and t.requires_grad for t in tensors)


class _WNConvBase(nn.Module):
    """Old-style torch.nn.utils.weight_norm parametrisation (dim=0): parameters weight_g, weight_v
    (+ bias) — the names the reference checkpoints carry (autoencoders.py:8, :23-27).

    Two parameter layouts, as in the reference: weight-normed (weight_g, weight_v; `module.weight` reads as the folded tensor)
    and, after torch.nn.utils.remove_weight_norm / remove_weight_norm_from_model or after loading a checkpoint that was saved
    that way, folded (one `weight` parameter).  load_state_dict accepts either layout into either form."""

    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        if self.transposed:
            wshape = (in_channels, out_channels, kernel_size)
        else:
            wshape = (out_channels, in_channels, kernel_size)
        # nn.Conv1d default init (kaiming_uniform a=sqrt(5)), then g = ||v|| as weight_norm does
        v = torch.empty(wshape)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.weight_v = nn.Parameter(v)
        if bias:
            # fan_in as torch computes it for the (possibly transposed) weight tensor: size(1) * K
            fan_in = wshape[1] * kernel_size
            bound = 1.0 / math.sqrt(fan_in)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self.register_forward_pre_hook(_NativeWeightNorm("weight", 0))
        self.__dict__["_derived"] = Fn.DerivedCache()

    # ---- the two parameter layouts ----
    @property
    def is_folded(self):
        return "weight" in self._parameters

    def __getattr__(self, name):
        # weight-normed form: `module.weight` is the folded tensor, as the attribute old-style weight_norm keeps on the module
        if name == "weight" and "weight_v" in self.__dict__.get("_parameters", ()):
            return self._fold_host()
        return super().__getattr__(name)

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = Fn.DerivedCache() if k == "_derived" else copy.deepcopy(v, memo)
        return new

    def _fold_host(self):
        """w = g * v / ||v|| in plain torch: parameter management on whatever device the module is on (attribute access, weight-norm
        removal, checkpoint conversion) — not the compute path, which folds inside the HIP fusion unit (folded_weight)."""
        v, g = self._parameters["weight_v"], self._parameters["weight_g"]
        return v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))

    def _to_folded(self, w):
        rg = self._parameters["weight_v"].requires_grad
        del self._parameters["weight_g"], self._parameters["weight_v"]
        self._parameters["weight"] = nn.Parameter(w.clone(), requires_grad=rg)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        kw, kg, kv = prefix + "weight", prefix + "weight_g", prefix + "weight_v"
        if not self.is_folded and kw in state_dict and kv not in state_dict:
            # a checkpoint saved after weight-norm removal: take the folded layout
            self._to_folded(torch.empty_like(self._parameters["weight_v"]))
        elif self.is_folded and kw not in state_dict and kv in state_dict and kg in state_dict:
            # weight-normed checkpoint into a module whose weight norm was removed ("pre_load" order of train.py:73-76): fold on load
            v, g = state_dict.pop(kv), state_dict.pop(kg)
            state_dict[kw] = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def p32(self, name, t):
        """fp32 view of a parameter for the kernels.  After `.half()` / `.bfloat16()` on the module (AutoencoderPretransform's
        `model_half`, models/pretransforms.py:48-49) the parameters are STORED in 16 bits — state_dict, memory and rounding as the
        reference — while the HIP conv stack keeps computing at fp32 accuracy on those rounded values: the widened copy is made once
        per parameter version (DerivedCache).  16-bit parameters cannot be trained through this path."""
        if t is None or t.dtype == torch.float32:
            return t
        if t.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("16-bit parameters are inference-only on the HIP conv stack (freeze the model: AutoencoderPretransform does)")
        return self._derived.get("f32/" + name, (t,), lambda: t.detach().float())

    def bias32(self):
        return self.p32("bias", self.bias)

    def folded_weight(self):
        """The conv weight for this pass: the `weight` parameter (folded layout), or g * v / ||v|| by sat_wn_fold — inside autograd
        when something is trained, from the layer's DerivedCache when nothing can ask for a gradient (no_grad, or a frozen layer)."""
        if self.is_folded:
            return self.p32("weight", self._parameters["weight"])
        v, g = self.p32("weight_v", self.weight_v), self.p32("weight_g", self.weight_g)
        if _inference_pass(v, g):
            return self._derived.get("folded", (v, g), lambda: Fn.WeightNormFn.apply(v.detach(), g.detach()))
        return Fn.WeightNormFn.apply(v, g)

    def weight_args(self):
        """(w, g) for the conv's autograd unit: (weight_v, weight_g) when the pass trains them — the unit folds and un-folds the weight
        norm itself (functional._wn_forward: one launch for dv / dg from the weight-gradient slabs) — else (folded weight, None)."""
        if not self.is_folded and Fn._ops(None).wn_fused:
            v, g = self.p32("weight_v", self.weight_v), self.p32("weight_g", self.weight_g)
            if not _inference_pass(v, g):
                return v, g
        return self.folded_weight(), None

    def derived_cache(self, *others):
        """The layer's DerivedCache when this call trains nothing of the layer (nor `others`: the SnakeBeta in front of it)."""
        ps = list(self._parameters.values()) + list(others)
        return self._derived if _inference_pass(*ps) else None


class WNConv1d(_WNConvBase):
    def forward(self, x, snake=None, res=None, tanh_out=False, next_snake=None):
        """next_snake: (log-alpha, log-beta, dilation) of the ResidualUnit that reads the output next — see ResidualUnit.forward."""
        a, b = (self.p32("alpha", snake.alpha), self.p32("beta", snake.beta)) if snake is not None else (None, None)
        if next_snake is not None and next_snake[0].dtype != torch.float32:
            next_snake = (self.p32("next_alpha", next_snake[0]), self.p32("next_beta", next_snake[1]), next_snake[2])
        w, g = self.weight_args()
        return Fn.SnakeConv1dFn.apply(x, a, b, w, self.bias32(), res, self.stride, self.dilation,
                                      self.padding, tanh_out, None, self.derived_cache(a, b), next_snake, g)


class WNConvTranspose1d(_WNConvBase):
    transposed = True

    def forward(self, x, snake=None):
        a, b = (self.p32("alpha", snake.alpha), self.p32("beta", snake.beta)) if snake is not None else (None, None)
        w, g = self.weight_args()
        return Fn.SnakeConvTr1dFn.apply(x, a, b, w, self.bias32(), self.stride, self.padding, None,
                                        self.derived_cache(a, b), g)


def _require_snake(use_snake, antialias_activation=False):
    if not use_snake:
        raise NotImplementedError("only use_snake=True (SnakeBeta, the Stable Audio Open / 2.0 VAE configuration) is "
                                  "implemented on the HIP path; the ELU variant is out of scope (SURVEY.md §8a V3)")
    if antialias_activation:
        raise NotImplementedError("antialias_activation (alias_free_torch) is out of scope")


class ResidualUnit(nn.Module):
    # activation recompute (the reference always checkpoints this unit in training, autoencoders.py:78-79): off by default — one
    # 47.55 s stereo item per GPU needs 31 GB here and 288 GB are available — switch on (class-wide or per instance) to halve the
    # unit's saved activations when the per-GPU batch does not fit; results are identical (tests/test_vae_parity.py)
    checkpointing = False
    # the unit as ONE launch (C <= 128: csrc/conv1d_bf16x3_k7q.h, FUSED).  None = automatic: fused without keeping the intermediate
    # under no_grad (inference), two launches when a backward will need it; True forces the fused launch (tests, measurements)
    fuse = None

    def __init__(self, in_channels, out_channels, dilation, use_snake=False, antialias_activation=False):
        super().__init__()
        _require_snake(use_snake, antialias_activation)
        self.dilation = dilation
        padding = (dilation * (7 - 1)) // 2
        self.layers = nn.Sequential(
            SnakeBeta(out_channels),
            WNConv1d(in_channels, out_channels, kernel_size=7, dilation=dilation, padding=padding),
            SnakeBeta(out_channels),
            WNConv1d(out_channels, out_channels, kernel_size=1),
        )

    def entry_snake(self):
        """(log-alpha, log-beta, dilation) of this unit's first activation + k7 conv: what the PRODUCER of its input needs to write
        the input as the k7 conv's activation planes in its own epilogue (csrc/conv1d_bf16x3.hip, plane emission)."""
        s1 = self.layers[0]
        return (s1.alpha, s1.beta, self.dilation)

    def forward(self, x, next_snake=None):
        """next_snake: entry_snake() of the ResidualUnit that reads this unit's output next (None: something else does)."""
        s1, c1, s2, c2 = self.layers
        ca, cb = c1.derived_cache(s1.alpha, s1.beta), c2.derived_cache(s2.alpha, s2.beta)
        if next_snake is not None and next_snake[0].dtype != torch.float32:
            next_snake = (c2.p32("next_alpha", next_snake[0]), c2.p32("next_beta", next_snake[1]), next_snake[2])
        (w1, g1), (w2, g2) = c1.weight_args(), c2.weight_args()
        return Fn.ResidualUnitFn.apply(x, c1.p32("alpha", s1.alpha), c1.p32("beta", s1.beta), w1, c1.bias32(),
                                       c2.p32("alpha", s2.alpha), c2.p32("beta", s2.beta), w2, c2.bias32(), self.dilation, None,
                                       self.checkpointing and torch.is_grad_enabled(), (ca, cb) if ca is not None and cb is not None else None,
                                       next_snake, self.fuse if self.fuse is not None else ("nokeep" if not torch.is_grad_enabled() else False),
                                       g1, g2)


class EncoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride, use_snake=False, antialias_activation=False):
        super().__init__()
        _require_snake(use_snake, antialias_activation)
        self.layers = nn.Sequential(
            ResidualUnit(in_channels, in_channels, dilation=1, use_snake=use_snake),
            ResidualUnit(in_channels, in_channels, dilation=3, use_snake=use_snake),
            ResidualUnit(in_channels, in_channels, dilation=9, use_snake=use_snake),
            SnakeBeta(in_channels),
            WNConv1d(in_channels, out_channels, kernel_size=2 * stride, stride=stride, padding=math.ceil(stride / 2)),
        )

    def forward(self, x, next_snake=None):
        """next_snake: entry_snake() of the first ResidualUnit of the NEXT block (it reads the down conv's output)."""
        r1, r2, r3, snake, down = self.layers
        x = r1(x, next_snake=r2.entry_snake())
        x = r2(x, next_snake=r3.entry_snake())
        return down(r3(x), snake=snake, next_snake=next_snake)


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride, use_snake=False, antialias_activation=False,
                 use_nearest_upsample=False):
        super().__init__()
        _require_snake(use_snake, antialias_activation)
        if use_nearest_upsample:
            raise NotImplementedError("use_nearest_upsample is out of scope (no in-repo Oobleck config enables it)")
        self.layers = nn.Sequential(
            SnakeBeta(in_channels),
            WNConvTranspose1d(in_channels, out_channels, kernel_size=2 * stride, stride=stride,
                              padding=math.ceil(stride / 2)),
            ResidualUnit(out_channels, out_channels, dilation=1, use_snake=use_snake),
            ResidualUnit(out_channels, out_channels, dilation=3, use_snake=use_snake),
            ResidualUnit(out_channels, out_channels, dilation=9, use_snake=use_snake),
        )

    def forward(self, x):
        snake, up, r1, r2, r3 = self.layers
        x = r1(up(x, snake=snake), next_snake=r2.entry_snake())       # (the transposed conv's depth-to-space epilogue emits no planes)
        return r3(r2(x, next_snake=r3.entry_snake()))


class OobleckEncoder(nn.Module):
    def __init__(self, in_channels=2, channels=128, latent_dim=32, c_mults=[1, 2, 4, 8], strides=[2, 4, 8, 8],
                 use_snake=False, antialias_activation=False):
        super().__init__()
        _require_snake(use_snake, antialias_activation)
        self.in_channels = in_channels
        c_mults = [1] + list(c_mults)
        self.depth = len(c_mults)
        layers = [WNConv1d(in_channels, c_mults[0] * channels, kernel_size=7, padding=3)]
        for i in range(self.depth - 1):
            layers += [EncoderBlock(c_mults[i] * channels, c_mults[i + 1] * channels, stride=strides[i], use_snake=use_snake)]
        layers += [SnakeBeta(c_mults[-1] * channels),
                   WNConv1d(c_mults[-1] * channels, latent_dim, kernel_size=3, padding=1)]
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        mods = list(self.layers)
        blocks = mods[1:-2]
        # the first conv (two audio channels in: csrc/edge_conv.hip) writes the first ResidualUnit's k7 activation planes beside its output
        x = mods[0](x, next_snake=blocks[0].layers[0].entry_snake() if blocks else None)
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1].layers[0].entry_snake() if i + 1 < len(blocks) else None
            x = blk(x, next_snake=nxt)
        return mods[-1](x, snake=mods[-2])


class OobleckDecoder(nn.Module):
    def __init__(self, out_channels=2, channels=128, latent_dim=32, c_mults=[1, 2, 4, 8], strides=[2, 4, 8, 8],
                 use_snake=False, antialias_activation=False, use_nearest_upsample=False, final_tanh=True):
        super().__init__()
        _require_snake(use_snake, antialias_activation)
        self.out_channels = out_channels
        c_mults = [1] + list(c_mults)
        self.depth = len(c_mults)
        layers = [WNConv1d(latent_dim, c_mults[-1] * channels, kernel_size=7, padding=3)]
        for i in range(self.depth - 1, 0, -1):
            layers += [DecoderBlock(c_mults[i] * channels, c_mults[i - 1] * channels, stride=strides[i - 1],
                                    use_snake=use_snake, antialias_activation=antialias_activation,
                                    use_nearest_upsample=use_nearest_upsample)]
        layers += [SnakeBeta(c_mults[0] * channels),
                   WNConv1d(c_mults[0] * channels, out_channels, kernel_size=7, padding=3, bias=False),
                   nn.Tanh() if final_tanh else nn.Identity()]
        self.final_tanh = final_tanh
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        mods = list(self.layers)
        x = mods[0](x)
        for blk in mods[1:-3]:
            x = blk(x)
        return mods[-2](x, snake=mods[-3], tanh_out=self.final_tanh)


class AudioAutoencoder(nn.Module):
    """Encoder / bottleneck / decoder orchestration (autoencoders.py:401-534, :601-732).
    Inner `pretransform` (wavelet/PQMF front-ends) is out of scope and must be None."""

    def __init__(self, encoder, decoder, latent_dim, downsampling_ratio, sample_rate, io_channels=2, bottleneck=None,
                 pretransform=None, in_channels=None, out_channels=None, soft_clip=False):
        super().__init__()
        if pretransform is not None:
            raise NotImplementedError("AudioAutoencoder(pretransform=...) is out of scope for the HIP path")
        self.downsampling_ratio = downsampling_ratio
        self.sample_rate = sample_rate
        self.latent_dim = latent_dim
        self.io_channels = io_channels
        self.in_channels = io_channels if in_channels is None else in_channels
        self.out_channels = io_channels if out_channels is None else out_channels
        self.min_length = self.downsampling_ratio
        self.bottleneck = bottleneck
        self.encoder = encoder
        self.decoder = decoder
        self.pretransform = None
        self.soft_clip = soft_clip
        self.is_discrete = self.bottleneck is not None and self.bottleneck.is_discrete

    def encode(self, audio, skip_bottleneck=False, return_info=False, skip_pretransform=False, iterate_batch=False, **kwargs):
        info = {}
        if iterate_batch:
            latents = torch.cat([self.encoder(audio[i:i + 1]) for i in range(audio.shape[0])], dim=0)
        else:
            latents = self.encoder(audio)
        info["pre_bottleneck_latents"] = latents
        if self.bottleneck is not None and not skip_bottleneck:
            latents, binfo = self.bottleneck.encode(latents, return_info=True, **kwargs)
            info.update(binfo)
        if return_info:
            return latents, info
        return latents

    def decode(self, latents, skip_bottleneck=False, iterate_batch=False, **kwargs):
        if self.bottleneck is not None and not skip_bottleneck:
            latents = self.bottleneck.decode(latents)
        if iterate_batch:
            decoded = torch.cat([self.decoder(latents[i:i + 1]) for i in range(latents.shape[0])], dim=0)
        else:
            decoded = self.decoder(latents, **kwargs)
        if self.soft_clip:
            decoded = torch.tanh(decoded)
        return decoded

    # -- chunked overlap-and-paste orchestration (autoencoders.py:601-732) --------------------
    def encode_audio(self, audio, chunked=False, overlap=32, chunk_size=128, **kwargs):
        """`overlap` / `chunk_size` are in latent frames (autoencoders.py:601-669)."""
        if not chunked:
            return self.encode(audio, **kwargs)
        r = int(self.downsampling_ratio)
        total = audio.shape[2]                      # samples; the grid is laid out in samples as the reference does (:622-635)
        csz, hop = chunk_size * r, (chunk_size - overlap) * r
        starts = list(range(0, total - csz + 1, hop))
        if not starts:
            raise ValueError("chunked encode needs at least chunk_size latent frames of audio")
        if starts[-1] + csz != total:
            starts.append(total - csz)              # final chunk = audio[..., -chunk_size:]  (:633)
        y_size = total // r
        out = torch.zeros((audio.shape[0], self.latent_dim, y_size), device=audio.device, dtype=audio.dtype)
        half = overlap // 2
        # noise= may be a list / tuple with one tensor per chunk (native extension: the reference draws each chunk's VAE noise from
        # torch's global generator, autoencoders.py:646 -> bottleneck.py:109; an explicit list makes a chunked encode reproducible on
        # any device); a single tensor is handed to every chunk as before
        chunk_noise = kwargs.pop("noise", None)
        if isinstance(chunk_noise, (list, tuple)) and len(chunk_noise) != len(starts):
            raise ValueError(f"chunked encode: {len(starts)} chunks but {len(chunk_noise)} noise tensors")
        for i, s0 in enumerate(starts):
            if chunk_noise is not None:
                kwargs["noise"] = chunk_noise[i] if isinstance(chunk_noise, (list, tuple)) else chunk_noise
            y = self.encode(audio[:, :, s0:s0 + csz], **kwargs)
            last = i == len(starts) - 1
            t1 = y_size if last else i * hop // r + chunk_size
            t0 = t1 - y.shape[2] if last else i * hop // r
            k0, k1 = 0, y.shape[2]
            if i > 0:
                t0, k0 = t0 + half, k0 + half
            if not last:
                t1, k1 = t1 - half, k1 - half
            out[:, :, t0:t1] = y[:, :, k0:k1]
        return out

    def decode_audio(self, latents, chunked=False, overlap=32, chunk_size=128, **kwargs):
        """Chunked decode: keep the centre of every chunk, last chunk right-aligned (autoencoders.py:671-732)."""
        if not chunked:
            return self.decode(latents, **kwargs)
        r = int(self.downsampling_ratio)
        out = torch.zeros((latents.shape[0], self.out_channels, latents.shape[2] * r), device=latents.device,
                          dtype=latents.dtype)
        for win in _chunk_windows(latents.shape[2], chunk_size, overlap):
            y = self.decode(latents[:, :, win.src0:win.src1])
            out[:, :, win.dst0 * r:win.dst1 * r] = y[:, :, win.keep0 * r:win.keep1 * r]
        return out


class _Win:
    __slots__ = ("src0", "src1", "dst0", "dst1", "keep0", "keep1")


def _chunk_windows(total, chunk, overlap):
    """Windows (all in latent frames) of the reference's overlap-and-paste scheme
    (autoencoders.py:626-668 / :689-731): chunks of `chunk` frames every `chunk - overlap` frames, a
    final right-aligned chunk if the grid does not end exactly at `total`; from every chunk the
    half-overlap at each interior edge is discarded."""
    hop = chunk - overlap
    starts = list(range(0, total - chunk + 1, hop))
    if not starts:
        raise ValueError("chunked encode/decode needs at least chunk_size latent frames")
    if starts[-1] + chunk != total:
        starts.append(total - chunk)
    half = overlap // 2
    wins = []
    for n, s0 in enumerate(starts):
        w = _Win()
        last = n == len(starts) - 1
        w.src0, w.src1 = s0, s0 + chunk
        # nominal destination of a grid chunk is n*hop; the last one is pasted at the very end
        d0 = (total - chunk) if last else n * hop
        w.keep0 = half if n > 0 else 0
        w.keep1 = chunk if last else chunk - half
        w.dst0, w.dst1 = d0 + w.keep0, d0 + w.keep1
        wins.append(w)
    return wins


def create_encoder_from_config(cfg):
    if cfg.get("type") != "oobleck":
        raise NotImplementedError(f"encoder type {cfg.get('type')!r} is out of scope (only 'oobleck' is on the hot path)")
    enc = OobleckEncoder(**cfg["config"])
    if not cfg.get("requires_grad", True):
        for p in enc.parameters():
            p.requires_grad = False
    return enc


def create_decoder_from_config(cfg):
    if cfg.get("type") != "oobleck":
        raise NotImplementedError(f"decoder type {cfg.get('type')!r} is out of scope (only 'oobleck' is on the hot path)")
    dec = OobleckDecoder(**cfg["config"])
    if not cfg.get("requires_grad", True):
        for p in dec.parameters():
            p.requires_grad = False
    return dec


def create_autoencoder_from_config(config):
    """Same JSON surface as autoencoders.py:867-910 for the in-scope types."""
    ae = config["model"]
    if ae.get("pretransform") is not None:
        raise NotImplementedError("autoencoder-level pretransform is out of scope")
    encoder = create_encoder_from_config(ae["encoder"])
    decoder = create_decoder_from_config(ae["decoder"])
    bottleneck = None
    if ae.get("bottleneck") is not None:
        if ae["bottleneck"]["type"] != "vae":
            raise NotImplementedError("only the 'vae' bottleneck is on the hot path")
        bottleneck = VAEBottleneck()
    return AudioAutoencoder(encoder, decoder, latent_dim=ae["latent_dim"], downsampling_ratio=ae["downsampling_ratio"],
                            sample_rate=config["sample_rate"], io_channels=ae["io_channels"], bottleneck=bottleneck,
                            in_channels=ae.get("in_channels"), out_channels=ae.get("out_channels"),
                            soft_clip=ae["decoder"].get("soft_clip", False))
