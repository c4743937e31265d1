"""Audio wire format against the independent numpy restatement in oracle/audio_front_end.py.
Tolerance 2e-3 absolute on the normalised dB scale (fp32 STFT against float64; one unit = 19.9 dB)."""
import numpy as np
import torch

import avformer_amd as A
from oracle.audio_front_end import mel_features


def _wave(seconds, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(int(44100 * seconds)) / 44100.0
    return 0.3 * torch.sin(2 * torch.pi * 440.0 * t) + 0.1 * torch.sin(2 * torch.pi * 3000.0 * t + 1.0) \
        + 0.02 * torch.randn(t.numel(), generator=g)


def test_full_length_clip_shape_and_values():
    fe = A.audio.MelFrontEnd()
    assert (fe.n_fft, fe.win_length, fe.hop_length, fe.full_frames) == (1024, 882, 441, 1001)
    x = _wave(10.0, 0)
    y = fe(x[None, None])                       # [B, 1, samples] -> [B, 1, 64, 1001]
    assert y.shape == (1, 1, 64, 1001) and y.dtype == torch.float32
    ref = mel_features(x.numpy())
    assert np.abs(y[0, 0].numpy() - ref).max() < 2e-3


def test_short_clip_is_left_padded_before_the_db_conversion():
    fe = A.audio.MelFrontEnd()
    x = _wave(1.3, 1)
    y = fe(x[None])                             # [1, samples] -> [1, 64, 1001]
    assert y.shape == (1, 64, 1001)
    ref = mel_features(x.numpy())
    assert np.abs(y[0].numpy() - ref).max() < 2e-3
    floor = y[0, :, :800]
    assert torch.allclose(floor, floor[0, 0].expand_as(floor))   # the padded frames sit on the top_db floor


def test_batch_elements_are_clamped_independently():
    fe = A.audio.MelFrontEnd()
    a, b = _wave(10.0, 2), 1e-3 * _wave(10.0, 3)
    y = fe(torch.stack([a, b])[:, None])        # [2, 1, samples]
    assert np.abs(y[0, 0].numpy() - mel_features(a.numpy())).max() < 2e-3
    assert np.abs(y[1, 0].numpy() - mel_features(b.numpy())).max() < 2e-3


def test_two_dimensional_batch_is_clamped_per_clip_too():
    """audio[B, samples] (no channel axis) took the top_db peak over the whole batch - a quiet clip batched
    with a loud one was clamped against the loud clip's peak, unlike the reference's per-clip AmplitudeToDB"""
    fe = A.audio.MelFrontEnd()
    a, b = _wave(2.0, 4), 1e-4 * _wave(2.0, 5)
    y = fe(torch.stack([a, b]))                 # [2, samples] -> [2, 64, 1001]
    assert y.shape == (2, 64, 1001)
    assert np.abs(y[0].numpy() - mel_features(a.numpy())).max() < 2e-3
    assert np.abs(y[1].numpy() - mel_features(b.numpy())).max() < 2e-3
    assert torch.equal(y[1], fe(b[None])[0])    # the same clip alone
    assert torch.equal(fe(a), y[0])             # and a bare [samples] clip
