"""Audio front-end wire format: waveform -> normalised log-mel spectrogram [.., 1, n_mels, frames],
the tensor the reference's data loader feeds the audio stream with (dataloader/aff2compdataset.py:47-68, 214-247;
dataloader/clip_transforms.py:59-108).

The reference builds it from ``torchaudio.transforms.MelSpectrogram`` + ``AmplitudeToDB('power', 80)`` + ``Normalize``.
torchaudio is not part of this image, so the published definitions of those transforms are restated here on plain torch
ops (``torch.stft`` runs on rocFFT on the GPU) - device-agnostic glue either side of the hot path, not a HIP kernel:

  * n_fft = 2^ceil(log2(window_size * sample_rate)) = 1024, win_length = 882, hop = 441, periodic Hann window, centred
    frames with reflect padding, one-sided power spectrum (|STFT|^2)                    (aff2compdataset.py:48-52, 60-65)
  * mel filterbank: HTK scale, f_min = 0, f_max = sample_rate / 2, triangular, un-normalised (torchaudio defaults)
  * clips shorter than ``sample_len_secs`` are LEFT-padded with zero frames BEFORE the dB conversion (235-239)
  * dB: 10 log10(max(x, 1e-10)), then clamped to (max over the clip) - 80                  (clip_transforms.py:96-108)
  * (x - mean) / std with mean = -14.8, std = 19.895                                         (aff2compdataset.py:67-68)

Parity: unpinned by the reference (its transform cannot be imported here); checked against an independent numpy
restatement (oracle/audio_front_end.py) and against a third-party implementation of the same published transform
(transformers.audio_utils; tests/test_audio_cpu.py).
"""
from __future__ import annotations

import math

import torch
from torch import nn


def melscale_fbanks_htk(n_freqs: int, n_mels: int, sample_rate: int, f_min: float = 0.0, f_max: float | None = None):
    """[n_freqs, n_mels] triangular filters on the HTK mel scale, no area normalisation."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (torch.pow(10.0, m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0).to(torch.float32)


class MelFrontEnd(nn.Module):
    """``forward(audio[..., samples]) -> features[..., n_mels, frames]`` (add the channel axis the caller's layout wants)."""

    def __init__(self, sample_rate: int = 44100, window_size: float = 20e-3, window_stride: float = 10e-3, n_mels: int = 64,
                 sample_len_secs: int = 10, top_db: float = 80.0, mean: float = -14.8, std: float = 19.895):
        super().__init__()
        self.sample_rate = sample_rate
        self.n_fft = 2 ** math.ceil(math.log2(window_size * sample_rate))
        self.win_length = int(window_size * sample_rate)
        self.hop_length = int(window_stride * sample_rate)
        self.n_mels = n_mels
        self.full_frames = int(sample_len_secs / window_stride + 1)
        self.top_db, self.mean, self.std = top_db, mean, std
        self.register_buffer("window", torch.hann_window(self.win_length), persistent=False)
        self.register_buffer("fb", melscale_fbanks_htk(self.n_fft // 2 + 1, n_mels, sample_rate), persistent=False)

    def mel_power(self, audio: torch.Tensor) -> torch.Tensor:
        lead = audio.shape[:-1]
        x = audio.reshape(-1, audio.shape[-1]).to(torch.float32)
        spec = torch.stft(x, self.n_fft, hop_length=self.hop_length, win_length=self.win_length, window=self.window,
                          center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        power = spec.real ** 2 + spec.imag ** 2                       # [b, n_freqs, frames]
        mel = torch.matmul(power.transpose(-1, -2), self.fb).transpose(-1, -2)
        return mel.reshape(*lead, self.n_mels, mel.shape[-1])

    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        mel = self.mel_power(audio)
        if mel.shape[-1] < self.full_frames:  # short clip: zero frames in front (aff2compdataset.py:235-239)
            pad = mel.new_zeros(*mel.shape[:-1], self.full_frames)
            pad[..., -mel.shape[-1]:] = mel
            mel = pad
        db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
        # top_db: relative to the maximum of EACH clip, as the reference's per-clip AmplitudeToDB on [1, n_mels, T]
        # (aff2compdataset.py:60-68): audio[samples] -> one clip; audio[B, samples] -> per row; audio[B, C, samples] ->
        # per row over its channels.  Never across the batch (a quiet clip batched with a loud one keeps its own floor).
        peak = db.amax(dim=(-3, -2, -1), keepdim=True) if db.dim() >= 4 else db.amax(dim=(-2, -1), keepdim=True)
        db = torch.maximum(db, peak - self.top_db)
        return (db - self.mean) / self.std
