"""Oracle: the three STFT_Process variants + audio prep.

TEST INFRASTRUCTURE -- CPU restatement in torch float32 (same op order as the reference so the
DFT tables are bit-identical: the reference evaluates cos/sin on UNREDUCED float32 angles).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# ref window registries: FSMN/STFT_Process.py:37-44 (v1, periodic defaults),
# DFSMN/*/STFT_Process.py:36-44 (v1b: 'bartlett' is mapped to hamming there),
# NVIDIA_.../STFT_Process.py:89-98 (v2: hann_sym / hann_sqrt / povey added).
_WINDOWS_V1 = {
    "bartlett": torch.bartlett_window,
    "blackman": torch.blackman_window,
    "hamming": torch.hamming_window,
    "hann": torch.hann_window,
    "kaiser": lambda L: torch.kaiser_window(L, periodic=True, beta=12.0),
}
_WINDOWS_V2 = {
    "bartlett": lambda L: torch.bartlett_window(L, periodic=True),
    "blackman": lambda L: torch.blackman_window(L, periodic=True),
    "hamming": lambda L: torch.hamming_window(L, periodic=True),
    "hann": lambda L: torch.hann_window(L, periodic=True),
    "hann_sym": lambda L: torch.hann_window(L, periodic=False),
    "hann_sqrt": lambda L: torch.hann_window(L, periodic=False).pow(0.5),
    "povey": lambda L: torch.hann_window(L, periodic=False).pow(0.85),
    "kaiser": lambda L: torch.kaiser_window(L, periodic=True, beta=12.0),
}


def padded_window(win_length, n_fft, window_type, variant="v2"):
    """Window of length n_fft, centre zero-padded (a1).
    ref: FSMN/STFT_Process.py:46-58; NVIDIA_.../STFT_Process.py:102-115."""
    if variant == "v1b":
        table = dict(_WINDOWS_V1)
        table["bartlett"] = torch.hamming_window      # ref quirk: DFSMN/*/STFT_Process.py:38
    elif variant == "v1":
        table = _WINDOWS_V1
    else:
        table = _WINDOWS_V2
    fn = table.get(window_type, torch.hann_window if variant != "v2" else _WINDOWS_V2["hann"])
    w = fn(win_length).float()
    if win_length == n_fft:
        return w
    if win_length < n_fft:
        left = (n_fft - win_length) // 2
        return F.pad(w, (left, n_fft - win_length - left))
    s = (win_length - n_fft) // 2
    return w[s:s + n_fft]


def dft_tables(n_fft, window, variant="v2"):
    """(cos·w, -sin·w) each [F, n_fft] with F = n_fft//2 + 1 (a2).
    v1/v1b: omega = 2*pi*f*t/n_fft   (ref: FSMN/STFT_Process.py:88-98)
    v2    : omega = (2*pi/n_fft)*f*t (ref: NVIDIA_.../STFT_Process.py:204-212)."""
    half = n_fft // 2
    t = torch.arange(n_fft, dtype=torch.float32).unsqueeze(0)
    f = torch.arange(half + 1, dtype=torch.float32).unsqueeze(1)
    if variant == "v2":
        omega = (2.0 * torch.pi / n_fft) * f * t
    else:
        omega = 2 * torch.pi * f * t / n_fft
    return torch.cos(omega) * window.unsqueeze(0), -torch.sin(omega) * window.unsqueeze(0)


def stft(x, cos_k, sin_k, hop, center_pad=True, pad_mode="constant"):
    """x [B,1,L] f32 -> (real, imag) each [B,F,T] (a3).
    ref: FSMN/STFT_Process.py:144-157; NVIDIA_.../STFT_Process.py:265-279."""
    n_fft = cos_k.shape[-1]
    half = n_fft // 2
    if center_pad:
        if pad_mode == "reflect":
            x = F.pad(x, (half, half), mode="reflect")
        else:
            x = F.pad(x, (half, half))
    real = F.conv1d(x, cos_k.unsqueeze(1), stride=hop)
    imag = F.conv1d(x, sin_k.unsqueeze(1), stride=hop)
    return real, imag


# ------------------------------------------------------------------ a4 audio prep flavours
def prep_fsmn(audio_i16):
    """float, remove the window mean, pre-emphasis 0.97 keeping x[0].
    ref: FSMN/Export_FSMN_VAD.py:76-79.  audio [B,1,L]; the mean is per clip (SURVEY hard part 5)."""
    a = audio_i16.float()
    a = a - a.mean(dim=-1, keepdim=True)
    return torch.cat([a[:, :, :1], a[:, :, 1:] - 0.97 * a[:, :, :-1]], dim=-1)


def prep_two_tap(audio_i16, scale, in_sample_rate=16000):
    """left-zero-padded 2-tap conv [-0.97*scale, scale], with the export's in-graph resample to 16 kHz when it was built for
    another IN_SAMPLE_RATE: F.interpolate(linear, align_corners=False, scale_factor = 1 / (in_rate / 16000)) BEFORE the conv
    for a higher input rate, AFTER it for a lower one.
    ref: Export_NVIDIA_MarbleNet_VAD.py:199-204,237-254 (scale = 1/32768);
         FireRedVAD/Export_FireRedVAD.py:396-400,431-449 (scale = 1)."""
    k = torch.tensor([[[-0.97 * scale, scale]]], dtype=torch.float32)
    a = audio_i16.float()
    rate_scale = in_sample_rate / 16000.0
    model_rate_scale = 1.0 / rate_scale
    if rate_scale > 1.0:
        a = F.interpolate(a, scale_factor=model_rate_scale, mode="linear", align_corners=False)
    a = F.conv1d(F.pad(a, (1, 0)), k)
    if rate_scale < 1.0:
        a = F.interpolate(a, scale_factor=model_rate_scale, mode="linear", align_corners=False)
    return a
