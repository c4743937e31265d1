"""Audio wire format on the device: ``audio.MelFrontEnd`` run on the MI355X (torch.stft
on rocFFT, the mel projection, the per-clip dB clamp) against the independent numpy restatement oracle/audio_front_end.py, on
a batch of clips of different loudness and length.  Parity stays UNPINNED by the reference (torchaudio is not importable:
the reference's own transform cannot produce a fixture); tolerance 2e-3 absolute on the normalised dB scale, as on the CPU."""
import numpy as np
import pytest
import torch

import avformer_amd as A
from oracle.audio_front_end import mel_features

pytestmark = pytest.mark.gpu


def _wave(seconds, seed, gain=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(int(44100 * seconds)) / 44100.0
    return gain * (0.3 * torch.sin(2 * torch.pi * 440.0 * t) + 0.1 * torch.sin(2 * torch.pi * (1000.0 + 500.0 * seed) * t + 1.0)
                   + 0.02 * torch.randn(t.numel(), generator=g))


def test_mel_front_end_on_device_matches_the_numpy_restatement():
    fe = A.audio.MelFrontEnd().cuda()
    clips = [_wave(10.0, 0), _wave(10.0, 1, 1e-3), _wave(10.0, 2, 30.0), _wave(10.0, 3, 1e-5)]  # loud, quiet, hot, near-silent
    x = torch.stack(clips)[:, None].cuda()              # [B, 1, samples]
    y = fe(x)
    assert y.is_cuda and y.shape == (4, 1, 64, 1001) and y.dtype == torch.float32
    for i, c in enumerate(clips):
        ref = mel_features(c.numpy())
        assert np.abs(y[i, 0].cpu().numpy() - ref).max() < 2e-3, i
    # clips are clamped against their OWN peak: each one alone gives the same rows as in the batch
    for i, c in enumerate(clips):
        alone = fe(c[None, None].cuda())
        assert (alone[0] - y[i]).abs().max().item() < 1e-4, i


def test_short_clip_on_device_is_left_padded():
    fe = A.audio.MelFrontEnd().cuda()
    c = _wave(1.3, 5)
    y = fe(c[None].cuda())                               # [1, samples] -> [1, 64, 1001]
    assert y.shape == (1, 64, 1001)
    assert np.abs(y[0].cpu().numpy() - mel_features(c.numpy())).max() < 2e-3
    floor = y[0, :, :800]
    assert torch.allclose(floor, floor[0, 0].expand_as(floor))


def test_device_and_host_results_agree():
    fe_h = A.audio.MelFrontEnd()
    fe_d = A.audio.MelFrontEnd().cuda()
    x = torch.stack([_wave(3.0, 7), _wave(3.0, 8, 0.05)])
    assert (fe_d(x.cuda()).cpu() - fe_h(x)).abs().max().item() < 2e-3
