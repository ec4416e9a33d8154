"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain torch) of the MS-STFT discriminator and its losses, consumed from a
reference-format state_dict.  Product code never imports this.

Reference: stable_audio_tools/models/encodec.py — DiscriminatorSTFT.forward :95-106 (Spectrogram -> cat(real, imag) ->
'b c w t -> b c t w' -> [NormConv2d + LeakyReLU(0.2)] x 5 -> conv_post), MultiScaleSTFTDiscriminator :108-138;
stable_audio_tools/models/discriminators.py — get_hinge_losses :13-16, EncodecDiscriminator.loss :31-63.

Third-party arithmetic on this path: torchaudio.transforms.Spectrogram (setup.py pins torchaudio>=2.0.2; not in the reference
tree, not installed here) — restated from its published algorithm: torch.stft(x, n_fft, hop, win_length, window=hann_window(win_length)
[periodic], center=False, onesided, return_complex) divided by window.pow(2).sum().sqrt() when normalized=True.  No torchaudio output
exists for that one transform in this environment; it is pinned against an independent source instead — the closed-form spectra of
bin-centred sinusoids under the periodic Hann window and a direct float64 evaluation of the definition
(tests/test_spectrogram_closed_form.py: this restatement to 1e-10, the native kernel to 2e-6) — which fixes the window, the
normalisation, the frame placement (center=False) and the sign / bin convention, i.e. everything the published algorithm specifies.
Everything downstream of it is pinned by tests/golden/disc_tiny.npz, produced by the reference's own classes with this same
restatement standing in for torchaudio.
"""
import torch
import torch.nn.functional as F


def spectrogram(x, n_fft, hop, win_length):
    """(B, C, T) -> complex (B, C, n_fft/2+1, frames): torchaudio Spectrogram(power=None, normalized=True, center=False)."""
    b, c, t = x.shape
    w = torch.hann_window(win_length, dtype=x.dtype, device=x.device)
    z = torch.stft(x.reshape(b * c, t), n_fft, hop, win_length, w, center=False, return_complex=True)
    z = z / w.pow(2.0).sum().sqrt()
    return z.reshape(b, c, z.shape[-2], z.shape[-1])


def _wn_conv2d(sd, prefix, x, dilation, padding):
    v, g = sd[prefix + "conv.weight_v"], sd[prefix + "conv.weight_g"]
    w = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1, 1)          # weight_norm over dims (1, 2, 3)  (encodec.py:25)
    return F.conv2d(x, w, sd[prefix + "conv.bias"], stride=1, dilation=dilation, padding=padding)


def discriminator_stft(sd, prefix, x, n_fft, hop, win_length, dilations=(1, 2, 4), kernel_size=(3, 9)):
    """encodec.py:95-106.  Returns (logits, [feature maps])."""
    z = spectrogram(x, n_fft, hop, win_length)
    z = torch.cat([z.real, z.imag], dim=1).permute(0, 1, 3, 2)     # 'b c w t -> b c t w'
    fmap = []
    kh, kw = kernel_size
    specs = [((1, 1), ((kh - 1) // 2, (kw - 1) // 2))]
    specs += [((d, 1), (((kh - 1) * d) // 2, (kw - 1) // 2)) for d in dilations]
    specs += [((1, 1), ((kh - 1) // 2, (kh - 1) // 2))]
    for i, (dil, pad) in enumerate(specs):
        z = F.leaky_relu(_wn_conv2d(sd, f"{prefix}convs.{i}.", z, dil, pad), 0.2)
        fmap.append(z)
    return _wn_conv2d(sd, prefix + "conv_post.", z, (1, 1), ((kh - 1) // 2, (kh - 1) // 2)), fmap


def ms_stft_discriminator(sd, x, n_ffts, hop_lengths, win_lengths, prefix="discriminators.discriminators."):
    outs = [discriminator_stft(sd, f"{prefix}{i}.", x, n, h, w) for i, (n, h, w) in enumerate(zip(n_ffts, hop_lengths, win_lengths))]
    return [o[0] for o in outs], [o[1] for o in outs]


def discriminator_losses(sd, reals, fakes, n_ffts, hop_lengths, win_lengths):
    """EncodecDiscriminator.loss (discriminators.py:31-63), hinge: (dis_loss, adv_loss, feature_matching_distance)."""
    lt, ft = ms_stft_discriminator(sd, reals, n_ffts, hop_lengths, win_lengths)
    lf, ff = ms_stft_discriminator(sd, fakes, n_ffts, hop_lengths, win_lengths)
    fm = dis = adv = 0.0
    for i in range(len(lt)):
        fm = fm + sum((a - b).abs().mean() for a, b in zip(ft[i], ff[i])) / len(ft[i])
        dis = dis + torch.relu(1 - lt[i]).mean() + torch.relu(1 + lf[i]).mean()
        adv = adv - lf[i].mean()
    n = len(lt)
    return dis / n, adv / n, fm / n
