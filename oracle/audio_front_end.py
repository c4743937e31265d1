"""Independent numpy restatement of the audio wire format (test infrastructure; see the package's audio.py for the
reference lines).  Frames are cut and windowed by hand and transformed with numpy.fft.rfft, the filterbank is built from
the HTK formulas directly: nothing is shared with the torch implementation under test."""
import math

import numpy as np


def mel_features(audio: np.ndarray, sample_rate=44100, window_size=20e-3, window_stride=10e-3, n_mels=64,
                 sample_len_secs=10, top_db=80.0, mean=-14.8, std=19.895) -> np.ndarray:
    """audio [samples] -> [n_mels, frames] float64"""
    n_fft = 2 ** math.ceil(math.log2(window_size * sample_rate))
    win = int(window_size * sample_rate)
    hop = int(window_stride * sample_rate)
    n = np.arange(win)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / win)          # periodic Hann
    left = (n_fft - win) // 2
    wpad = np.zeros(n_fft)
    wpad[left:left + win] = w
    x = np.pad(audio.astype(np.float64), (n_fft // 2, n_fft // 2), mode="reflect")
    frames = 1 + (len(x) - n_fft) // hop
    power = np.empty((n_fft // 2 + 1, frames))
    for t in range(frames):
        power[:, t] = np.abs(np.fft.rfft(x[t * hop:t * hop + n_fft] * wpad)) ** 2
    freqs = np.linspace(0, sample_rate // 2, n_fft // 2 + 1)
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(sample_rate // 2), n_mels + 2))
    fb = np.zeros((len(freqs), n_mels))
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        fb[:, m] = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)))
    mel = fb.T @ power
    full = int(sample_len_secs / window_stride + 1)
    if mel.shape[1] < full:
        mel = np.concatenate([np.zeros((n_mels, full - mel.shape[1])), mel], axis=1)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - top_db)
    return (db - mean) / std
