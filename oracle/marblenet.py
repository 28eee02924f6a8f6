"""Oracle: NVIDIA Frame-VAD MarbleNet v2.0 path.

TEST INFRASTRUCTURE -- CPU restatement in torch float32.

In tree (pinned): BN folding `fold_bn_into_conv1d` (Export_NVIDIA_MarbleNet_VAD.py:58-105), the
wrapper front-end (:236-262), softmax/split/`signal_len - 1` (:265-275), the host loop
(Inference_NVIDIA_MarbleNet_VAD_ONNX.py:130-147,369-402).
NOT in tree (NeMo `EncDecFrameClassificationModel`, un-vendored, version unpinned => PARITY UNPINNED):
the ConvASREncoder / decoder themselves.  Restated from the published NeMo config
`marblenet_3x2x64_20ms` (Jasper blocks, separable, masks off as the reference forces at :210-216):
    B1  filters 128, repeat 1, kernel 11, stride 2,               no residual
    B2  filters  64, repeat 2, kernel 13,                         residual (1x1 conv + BN of block input)
    B3  filters  64, repeat 2, kernel 15,                         residual
    B4  filters  64, repeat 2, kernel 17,                         residual
    B5  filters 128, repeat 1, kernel 29, dilation 2,             no residual
    B6  filters 128, repeat 1, kernel 1 (plain conv),             no residual
    each sub-block: depthwise conv (same padding) -> pointwise 1x1 -> BatchNorm -> ReLU
    decoder: Linear(128 -> 2); wrapper applies softmax.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mel as omel
from . import postproc
from . import stft as ostft

BLOCKS = (  # (filters, repeat, kernel, stride, dilation, residual, separable)
    (128, 1, 11, 2, 1, False, True),
    (64, 2, 13, 1, 1, True, True),
    (64, 2, 15, 1, 1, True, True),
    (64, 2, 17, 1, 1, True, True),
    (128, 1, 29, 1, 2, False, True),
    (128, 1, 1, 1, 1, False, False),
)
EPS = 1e-3     # NeMo Jasper BatchNorm1d eps


def fold_bn(w, b, gamma, beta, mean, var, eps):
    """ref: fold_bn_into_conv1d, Export_NVIDIA_MarbleNet_VAD.py:58-105."""
    scale = gamma * torch.rsqrt(var + eps)
    new_w = w * scale.reshape(-1, 1, 1)
    new_b = ((b - mean) if b is not None else (-mean)) * scale + beta
    return new_w, new_b


class Frontend:
    def __init__(self):
        win = ostft.padded_window(400, 512, "hann_sym", "v2")
        self.cos_k, self.sin_k = ostft.dft_tables(512, win, "v2")
        self.fbank = omel.melscale_fbanks(257, 0, 8000, 80, 16000, "slaney", "slaney").t().unsqueeze(0)


def log_mel(fe, audio_i16, in_sample_rate=16000):
    """int16 [B,1,L] -> [B,80,T], T = L'//160+1 (L' = length after the in-graph resample). ref: Export_NVIDIA_MarbleNet_VAD.py:236-262."""
    a = ostft.prep_two_tap(audio_i16, 1.0 / 32768.0, in_sample_rate)
    re, im = ostft.stft(a, fe.cos_k, fe.sin_k, 160, True)
    return omel.log_mel(re, im, fe.fbank, 1e-7, "add")


def _bn(x, w, p):
    return F.batch_norm(x, w[p + "_mean"], w[p + "_var"], w[p + "_gamma"], w[p + "_beta"], False, 0.0, EPS)


def encoder(w, x):
    """[B,80,T] -> ([B,128,T'], lengths).  Unfolded weights (conv + BatchNorm in eval mode)."""
    length = x.shape[-1]
    cin = x.shape[1]
    for bi, (filt, rep, k, stride, dil, residual, sep) in enumerate(BLOCKS):
        block_in = x
        for r in range(rep):
            p = f"b{bi}r{r}"
            if sep:
                pad = (dil * (k - 1)) // 2
                x = F.conv1d(x, w[p + "_dw"].unsqueeze(1), stride=stride, padding=pad, dilation=dil, groups=cin)
                length = (length + 2 * pad - dil * (k - 1) - 1) // stride + 1
                x = F.conv1d(x, w[p + "_pw"].unsqueeze(-1))
            else:
                x = F.conv1d(x, w[p + "_pw"].unsqueeze(-1))
            x = _bn(x, w, p)
            cin = filt
            if r < rep - 1:
                x = F.relu(x)
        if residual:
            res = _bn(F.conv1d(block_in, w[f"b{bi}res_pw"].unsqueeze(-1)), w, f"b{bi}res")
            x = x + res
        x = F.relu(x)
    return x, length


def forward(fe, w, audio_i16, in_sample_rate=16000):
    """session.run equivalent: int16 [B,1,L] -> (score_silence, score_active [B,T',1], signal_len-1)."""
    enc, length = encoder(w, log_mel(fe, audio_i16, in_sample_rate))
    logits = F.linear(enc.transpose(1, 2), w["dec_w"], w["dec_b"])
    score = torch.softmax(logits, dim=-1)
    return score[..., :1], score[..., 1:], length - 1


def run_clip(fe, w, audio_i16_1d, post=(3, 0.5, 10, 1000, 10, 3, 0), frame_shift_s=0.02, window=None, pad_noise=None):
    """Whole-clip driver.  window None = the dynamic-axis mode (one window = whole clip, <= 3600 s); an integer = a
    static-shape export: non-overlapping windows on a noise-padded grid, each window's first min(signal_len, T) frames
    concatenated.  ref: Inference_NVIDIA_MarbleNet_VAD_ONNX.py:130-147 (grid), :369-388 (loop), :391-402 (post).
    `pad_noise` replaces the reference's unseeded np.random.normal."""
    a = np.asarray(audio_i16_1d)
    n = a.shape[0]
    L = min(16000 * 3600, n) if window is None else int(window)
    padded, _ = postproc.pad_to_window_grid(a, L, L, pad_noise)
    chunks = []
    s = 0
    while s + L <= padded.shape[0]:
        sil, act, slen = forward(fe, w, torch.from_numpy(padded[s:s + L].copy()).reshape(1, 1, -1))
        valid = min(int(slen), act.shape[1])
        chunks.append(act[0, :valid, 0].numpy())
        s += L
    probs = np.concatenate(chunks, axis=0) if chunks else np.zeros((0,), dtype=np.float32)
    pp = postproc.VadPostprocessor(*post, frame_shift_s=frame_shift_s, frame_length_s=None)
    dec = pp.process(probs)
    return pp.decision_to_segment(dec, n / 16000), probs, dec
