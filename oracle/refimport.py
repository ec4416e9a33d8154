"""TEST INFRASTRUCTURE ONLY — imports the *reference* itself.

Where it comes from, in this order: $SAT_REFERENCE_ROOT, the read-only checkout /root/reference (build container), or
the copy `oracle/stage_ref.py` stages into the git-ignored `oracle/_ref/` (what the GPU box sees: /root/reference does
not exist there, the staged tree travels with the working tree like the built .so).  Users: `oracle/gen_golden*.py`,
the drop-in / pinning tests, and bench.py's `cpu_baseline` leg.  Nothing in the product package imports it.

The reference's hot-path modules import four off-path third-party packages at module scope
(SURVEY.md §8c).  They are stubbed in ``sys.modules`` so that the in-scope code imports unchanged:
  torchaudio(.transforms.Resample)  — autoencoders.py:9, pretransforms.py:4
  alias_free_torch.Activation1d     — autoencoders.py:10 (only used if antialias_activation)
  k_diffusion                       — inference/sampling.py:6 (only used by sample_k)
  einops_exts.rearrange_many        — adp.py:14
"""
import importlib.util
import os
import sys
import types

_STAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _find_root():
    for cand in (os.environ.get("SAT_REFERENCE_ROOT"), "/root/reference", _STAGED):
        if cand and os.path.isdir(os.path.join(cand, "stable_audio_tools")):
            return cand
    return "/root/reference"


REF_ROOT = _find_root()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch

    class _Unavailable(torch.nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("stubbed third-party module (off the hot path)")

    class _Spectrogram(torch.nn.Module):
        """torchaudio.transforms.Spectrogram restated from its published algorithm for the one call site on the scoped path
        (models/encodec.py:73-76: power=None, normalized=True, center=False, window_fn=torch.hann_window).  PARITY UNPINNED:
        torchaudio itself is absent here (oracle/disc_oracle.py header)."""

        def __init__(self, n_fft, hop_length, win_length, window_fn=torch.hann_window, normalized=False, center=True, pad_mode="reflect",
                     power=2.0):
            super().__init__()
            assert power is None and not center
            self.n_fft, self.hop_length, self.win_length, self.normalized = n_fft, hop_length, win_length, normalized
            self.register_buffer("window", window_fn(win_length), persistent=False)

        def forward(self, x):
            shp = x.shape
            z = torch.stft(x.reshape(-1, shp[-1]), self.n_fft, self.hop_length, self.win_length, self.window.to(x), center=False,
                           return_complex=True)
            if self.normalized:
                z = z / self.window.to(x).pow(2.0).sum().sqrt()
            return z.reshape(*shp[:-1], z.shape[-2], z.shape[-1])

    if "torchaudio" not in sys.modules:
        ta = _stub("torchaudio")
        ta.transforms = _stub("torchaudio.transforms", Resample=_Unavailable, Spectrogram=_Spectrogram)
        ta.functional = _stub("torchaudio.functional")
    if "alias_free_torch" not in sys.modules:
        _stub("alias_free_torch", Activation1d=_Unavailable)
    if "k_diffusion" not in sys.modules:
        kd = _stub("k_diffusion")
        kd.external = _stub("k_diffusion.external")
        kd.sampling = _stub("k_diffusion.sampling")
    if "einops_exts" not in sys.modules:
        _stub("einops_exts", rearrange_many=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub")))


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "stable_audio_tools"))


def import_reference():
    """Returns the reference `stable_audio_tools` package (models only)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REF_ROOT}")
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import stable_audio_tools  # noqa: F401
    return stable_audio_tools


def import_auraloss():
    """training/losses/auraloss.py must be loaded by file path: the package __init__ chain pulls in
    torchaudio / pytorch_lightning."""
    path = os.path.join(REF_ROOT, "stable_audio_tools", "training", "losses", "auraloss.py")
    spec = importlib.util.spec_from_file_location("ref_auraloss", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
