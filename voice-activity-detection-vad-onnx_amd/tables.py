"""Constant tables of the signal front-end, built on the host at load time (init only).

The DFT tables MUST be the reference's own float32 values -- it evaluates cos/sin on unreduced
float32 angles (up to ~1600 rad), so its basis differs from an exact DFT by up to 8e-5 per entry
(SURVEY hard part 1).  We therefore build them with torch-CPU float32 ops in the same order as
the reference (`STFT_Process.__init__`: FSMN/STFT_Process.py:87-98 for v1/v1b,
NVIDIA_.../STFT_Process.py:204-217 for v2) and hand them to the HIP kernel as data.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def analysis_window(kind, win_length, n_fft, variant):
    """Centre-zero-padded window (create_padded_window: FSMN/STFT_Process.py:46-58,
    DFSMN/*/STFT_Process.py:47-63, NVIDIA_.../STFT_Process.py:102-115)."""
    periodic = {"hann": torch.hann_window, "hamming": torch.hamming_window,
                "blackman": torch.blackman_window, "bartlett": torch.bartlett_window}
    if variant == "v1b" and kind == "bartlett":
        kind = "hamming"                      # the DFSMN copy maps 'bartlett' to hamming (line 38)
    if kind in periodic:
        w = periodic[kind](win_length, periodic=True)
    elif kind == "kaiser":
        w = torch.kaiser_window(win_length, periodic=True, beta=12.0)
    elif variant == "v2" and kind in ("hann_sym", "hann_sqrt", "povey"):
        w = torch.hann_window(win_length, periodic=False)
        if kind != "hann_sym":
            w = w.pow(0.5 if kind == "hann_sqrt" else 0.85)
    else:
        w = torch.hann_window(win_length, periodic=True)
    w = w.float()
    if win_length < n_fft:
        lead = (n_fft - win_length) // 2
        w = torch.cat([torch.zeros(lead), w, torch.zeros(n_fft - win_length - lead)])
    elif win_length > n_fft:
        s = (win_length - n_fft) // 2
        w = w[s:s + n_fft]
    return w


def windowed_dft(n_fft, window, variant):
    """(cos*w, -sin*w), float32 [n_fft//2+1, n_fft]; angle op order per variant."""
    t = torch.arange(n_fft, dtype=torch.float32).unsqueeze(0)
    f = torch.arange(n_fft // 2 + 1, dtype=torch.float32).unsqueeze(1)
    omega = (2.0 * torch.pi / n_fft) * f * t if variant == "v2" else 2 * torch.pi * f * t / n_fft
    return (torch.cos(omega) * window.unsqueeze(0)).contiguous(), (-torch.sin(omega) * window.unsqueeze(0)).contiguous()


def mel_filters_torchaudio(n_freqs, f_min, f_max, n_mels, sample_rate, norm, mel_scale):
    """torchaudio.functional.melscale_fbanks semantics -> [n_mels, n_freqs] float32
    (call sites: FSMN/Export_FSMN_VAD.py:63 htk/no norm; Export_NVIDIA_MarbleNet_VAD.py:186
    slaney/slaney; DFSMN/.../Export_DFSMN_VAD.py:308 htk)."""
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0

    def to_mel(hz):
        if mel_scale == "htk":
            return 2595.0 * math.log10(1.0 + (hz / 700.0))
        if hz >= min_log_hz:
            return min_log_mel + math.log(hz / min_log_hz) / logstep
        return hz / f_sp

    grid = torch.linspace(to_mel(float(f_min)), to_mel(float(f_max)), n_mels + 2)
    if mel_scale == "htk":
        edges = 700.0 * (10.0 ** (grid / 2595.0) - 1.0)
    else:
        edges = f_sp * grid
        hi = grid >= min_log_mel
        edges[hi] = min_log_hz * torch.exp(logstep * (grid[hi] - min_log_mel))
    bins = torch.linspace(0, sample_rate // 2, n_freqs)
    width = edges[1:] - edges[:-1]
    dist = edges.unsqueeze(0) - bins.unsqueeze(1)                 # [n_freqs, n_mels+2]
    rising = (-1.0 * dist[:, :-2]) / width[:-1]
    falling = dist[:, 2:] / width[1:]
    fb = torch.clamp(torch.minimum(rising, falling), min=0.0)
    if norm == "slaney":
        fb = fb * (2.0 / (edges[2:n_mels + 2] - edges[:n_mels])).unsqueeze(0)
    return fb.t().contiguous()


def mel_filters_kaldi(n_fft, n_mels, sample_rate, low_freq=20.0, high_freq=0.0):
    """In-tree Kaldi-style bank of FireRedVAD (Export_FireRedVAD.py:122-169) -> [n_mels, n_fft//2+1]."""
    if high_freq <= 0:
        high_freq = sample_rate / 2.0 + high_freq
    lg2 = math.log(2.0)

    def fwd(f):
        return f if f < 1000.0 else 1000.0 + 1000.0 * math.log(f / 1000.0) / lg2

    def inv(m):
        return m if m < 1000.0 else 1000.0 * math.exp((m - 1000.0) * lg2 / 1000.0)

    nb = n_fft // 2 + 1
    pts = torch.linspace(fwd(low_freq), fwd(high_freq), n_mels + 2)
    hz = torch.tensor([inv(p.item()) for p in pts], dtype=torch.float32)
    freqs = torch.linspace(0, sample_rate / 2.0, nb)
    fb = torch.zeros(n_mels, nb, dtype=torch.float32)
    for m in range(n_mels):
        a, b, c = hz[m], hz[m + 1], hz[m + 2]
        up = (freqs >= a) & (freqs <= b) & bool(b > a)
        dn = (freqs > b) & (freqs <= c) & bool(c > b)
        fb[m, up] = (freqs[up] - a) / (b - a)
        fb[m, dn] = (c - freqs[dn]) / (c - b)
    return fb


def as_np(t):
    return np.ascontiguousarray(t.numpy(), dtype=np.float32)
