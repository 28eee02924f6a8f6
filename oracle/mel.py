"""Oracle: mel filterbanks + log.

TEST INFRASTRUCTURE.  `melscale_fbanks` restates the public torchaudio algorithm
(torchaudio.functional.melscale_fbanks -- un-vendored third-party dependency of the reference,
version unpinned => PARITY UNPINNED for the filter values; call sites:
FSMN/Export_FSMN_VAD.py:63, Export_NVIDIA_MarbleNet_VAD.py:186, DFSMN/.../Export_DFSMN_VAD.py:308).
`kaldi_mel_filterbank` follows the in-tree FireRedVAD/Export_FireRedVAD.py:122-169 and is pinned.
"""
from __future__ import annotations

import math

import torch


def _hz_to_mel(freq, mel_scale):
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + (freq / 700.0))
    # slaney: linear below 1 kHz, log above
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    if freq >= min_log_hz:
        mels = min_log_hz / f_sp + math.log(freq / min_log_hz) / (math.log(6.4) / 27.0)
    return mels


def _mel_to_hz(mels, mel_scale):
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = 1000.0 * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """Triangular filterbank [n_freqs, n_mels], float32 (torchaudio semantics)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel(float(f_min), mel_scale)
    m_max = _hz_to_mel(float(f_max), mel_scale)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = _mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    if norm == "slaney":
        enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
        fb = fb * enorm.unsqueeze(0)
    return fb


def kaldi_mel_filterbank(n_fft, n_mels, sample_rate, low_freq=20.0, high_freq=0.0):
    """Kaldi-style mel (linear < 1 kHz, log2 above) triangles [n_mels, n_fft//2+1].
    ref: FireRedVAD/Export_FireRedVAD.py:122-169."""
    if high_freq <= 0:
        high_freq = sample_rate / 2.0 + high_freq

    def to_mel(f):
        return f if f < 1000.0 else 1000.0 + 1000.0 * math.log(f / 1000.0) / math.log(2.0)

    def from_mel(m):
        return m if m < 1000.0 else 1000.0 * math.exp((m - 1000.0) * math.log(2.0) / 1000.0)

    bins = n_fft // 2 + 1
    centers_mel = torch.linspace(to_mel(low_freq), to_mel(high_freq), n_mels + 2)
    hz = torch.tensor([from_mel(m.item()) for m in centers_mel], dtype=torch.float32)
    freqs = torch.linspace(0, sample_rate / 2.0, bins)
    fb = torch.zeros(n_mels, bins, dtype=torch.float32)
    for i in range(n_mels):
        lo, mid, hi = hz[i], hz[i + 1], hz[i + 2]
        for j in range(bins):
            fr = freqs[j]
            if lo <= fr <= mid and mid > lo:
                fb[i, j] = (fr - lo) / (mid - lo)
            elif mid < fr <= hi and hi > mid:
                fb[i, j] = (hi - fr) / (hi - mid)
    return fb


def log_mel(real, imag, fbank, floor, mode):
    """[B,F,T] x2 -> [B,n_mels,T]; mode 'clamp' = clamp(min=floor).log(), 'add' = (x+floor).log().
    ref: FSMN/Export_FSMN_VAD.py:81; Export_NVIDIA_MarbleNet_VAD.py:260-262;
         FireRedVAD/Export_FireRedVAD.py:455-461."""
    power = real * real + imag * imag
    mel = torch.matmul(fbank, power)
    if mode == "add":
        return (mel + floor).log()
    return mel.clamp(min=floor).log()
