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


def test_mel_power_against_a_third_party_implementation():
    """The reference's transform is torchaudio.transforms.MelSpectrogram (torchaudio 0.6, README.md:12), which this image does not
    hold - so no fixture can come from the reference itself and the front-end stays 'parity unpinned'.  What CAN be checked here:
    an implementation written by somebody else.  `transformers.audio_utils` (Hugging Face, an installed third-party package that
    documents its `spectrogram` / `mel_filter_bank` as reproducing torchaudio's MelSpectrogram with mel_scale='htk', norm=None)
    computes the same mel power spectrogram from the same parameters (aff2compdataset.py:48-65): n_fft 1024, win 882 (periodic
    Hann, centred in the frame), hop 441, reflect padding, power 2, 64 HTK mel filters over 0 .. 22050 Hz."""
    tau = __import__("pytest").importorskip("transformers.audio_utils")
    fe = A.audio.MelFrontEnd()
    x = _wave(2.0, 7)
    ours = fe.mel_power(x).double().numpy()                     # [64, frames]
    window = tau.window_function(fe.win_length, "hann", periodic=True, frame_length=fe.n_fft)
    filters = tau.mel_filter_bank(fe.n_fft // 2 + 1, fe.n_mels, 0.0, 22050.0, fe.sample_rate, norm=None, mel_scale="htk")
    theirs = tau.spectrogram(x.double().numpy(), window, frame_length=fe.n_fft, hop_length=fe.hop_length, fft_length=fe.n_fft,
                             power=2.0, center=True, pad_mode="reflect", onesided=True, mel_filters=filters, mel_floor=0.0)
    assert theirs.shape == ours.shape
    assert np.allclose(filters.T, fe.fb.double().numpy().T, atol=1e-6)          # the filterbank itself, filter by filter
    scale = np.abs(theirs).max()
    assert np.abs(ours - theirs).max() / scale < 2e-5            # fp32 STFT against float64
    # ... and on the dB scale the network sees (10 log10, floor 1e-10)
    db_o, db_t = 10 * np.log10(np.maximum(ours, 1e-10)), 10 * np.log10(np.maximum(theirs, 1e-10))
    keep = theirs > 1e-6 * scale                                 # (bins at the numerical floor differ in their noise, not in dB that matter)
    assert np.abs(db_o - db_t)[keep].max() < 0.02
