"""Oracle: FireRedVAD non-stream path (SURVEY §8 rows a17, a21 + front-end a3-a5).

TEST INFRASTRUCTURE -- CPU restatement in torch float32, batched over clips.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mel as omel
from . import postproc
from . import stft as ostft

NFFT, WIN, HOP, NMELS, SR = 400, 400, 160, 80, 16000


class Frontend:
    """ref: FireRedVAD/Export_FireRedVAD.py:379-418 (povey 400/400/160, no centre pad, Kaldi mel)."""

    def __init__(self):
        win = ostft.padded_window(WIN, NFFT, "povey", "v2")
        self.cos_k, self.sin_k = ostft.dft_tables(NFFT, win, "v2")
        self.fbank = omel.kaldi_mel_filterbank(NFFT, NMELS, SR, 20.0, 0.0).unsqueeze(0)   # [1,80,201]


def log_mel(fe, audio_i16):
    """int16 [B,1,L] -> [B,80,T], T = (L-400)//160+1. ref: Export_FireRedVAD.py:428-461."""
    a = ostft.prep_two_tap(audio_i16, 1.0)
    re, im = ostft.stft(a, fe.cos_k, fe.sin_k, HOP, center_pad=False)
    return omel.log_mel(re, im, fe.fbank, 1e-7, "clamp")


def fsmn_memory(x, lb_w, la_w, N1, S1, N2, S2):
    """x + causal depthwise FIR (+ strictly-future FIR). x [B,P,T].
    ref: FireRedVAD/Export_FireRedVAD.py:213-236."""
    P = x.shape[1]
    mem = x + F.conv1d(F.pad(x, ((N1 - 1) * S1, 0)), lb_w.unsqueeze(1), dilation=S1, groups=P)
    if N2 > 0 and x.size(2) > 1:
        la = F.conv1d(F.pad(x, (0, N2 * S2)), la_w.unsqueeze(1), dilation=S2, groups=P)
        mem = mem + la[:, :, S2:]
    return mem


def detect_model(w, feat):
    """DFSMN stack -> sigmoid probs [B,odim,T]. ref: Export_FireRedVAD.py:266-326.
    `w['cfg']` = dict(R,M,H,P,N1,S1,N2,S2,odim)."""
    c = w["cfg"]
    N1, S1, N2, S2 = c["N1"], c["S1"], c["N2"], c["S2"]

    def pw(x, wk, bk=None):          # Conv1d(kernel=1) == per-frame linear
        return F.conv1d(x, w[wk].unsqueeze(-1), None if bk is None else w[bk])

    h = F.relu(pw(feat, "fc1_w", "fc1_b"))
    p = F.relu(pw(h, "fc2_w", "fc2_b"))
    mem = fsmn_memory(p, w["fsmn0_lb"], w.get("fsmn0_la"), N1, S1, N2, S2)
    for r in range(1, c["R"]):
        hh = F.relu(pw(mem, f"blk{r}_fc1_w", f"blk{r}_fc1_b"))
        pp = pw(hh, f"blk{r}_fc2_w")
        mem = fsmn_memory(pp, w[f"fsmn{r}_lb"], w.get(f"fsmn{r}_la"), N1, S1, N2, S2) + mem
    x = mem
    for m in range(c["M"]):
        x = F.relu(pw(x, f"dnn{m}_w", f"dnn{m}_b"))
    return torch.sigmoid(pw(x, "out_w", "out_b"))


def forward(fe, w, audio_i16):
    """session.run equivalent: int16 [B,1,16000] -> probs [B,odim,98]. ref: Export_FireRedVAD.py:420-467."""
    return detect_model(w, log_mel(fe, audio_i16))


def valid_frame_count(num_samples):
    """ref: FireRedVAD/Inference_FireRed_ONNX.py:84-89 (IN_SAMPLE_RATE == 16000)."""
    return 0 if num_samples < WIN else 1 + (num_samples - WIN) // HOP


def run_clip(fe, w, audio_i16_1d, pad_noise, window=16000, post=(5, 0.4, 20, 2000, 20, 5, 0)):
    """Whole-clip driver for ONE clip: non-overlapping stateless windows, concat, truncate,
    VadPostprocessor, segments.  ref: FireRedVAD/Inference_FireRed_ONNX.py:535-591."""
    n = int(np.asarray(audio_i16_1d).shape[0])
    audio, _ = postproc.pad_to_window_grid(audio_i16_1d, window, window, pad_noise)
    probs = []
    for s in range(0, audio.shape[0] - window + 1, window):
        chunk = torch.from_numpy(audio[s:s + window].copy()).reshape(1, 1, -1)
        probs.append(forward(fe, w, chunk)[0, 0].numpy())
    allp = np.concatenate(probs, axis=0)[:valid_frame_count(n)] if probs else np.zeros((0,), np.float32)
    pp = postproc.VadPostprocessor(*post)
    dec = pp.process(allp)
    return pp.decision_to_segment(dec, n / SR), allp, dec
