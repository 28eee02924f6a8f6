"""Oracle: FSMN-VAD path.

TEST INFRASTRUCTURE -- CPU restatement in torch float32, batched over clips (the reference is
batch-1; every whole-tensor reduction there becomes a per-clip reduction here, SURVEY hard part 5).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import mel as omel
from . import postproc
from . import stft as ostft

NFFT, WIN, HOP, NMELS, SR = 512, 400, 160, 80, 16000
LFR_M, LFR_N = 5, 1
LORDER = 20
PROJ = 128


class Frontend:
    """Constant tables of FSMN_VAD.__init__. ref: FSMN/Export_FSMN_VAD.py:57-72,106."""

    def __init__(self, input_audio_len=16000):
        self.L = input_audio_len
        self.T = input_audio_len // HOP + 1
        win = ostft.padded_window(WIN, NFFT, "hamming", "v1")
        self.cos_k, self.sin_k = ostft.dft_tables(NFFT, win, "v1")
        self.fbank = omel.melscale_fbanks(NFFT // 2 + 1, 20, SR // 2, NMELS, SR, None, "htk").t().unsqueeze(0)
        self.lfr_half = (LFR_M - 1) // 2
        self.T_lfr = (self.T + LFR_N - 1) // LFR_N
        idx = torch.arange(0, self.T_lfr * LFR_N, LFR_N, dtype=torch.int64).unsqueeze(1) + torch.arange(LFR_M)
        self.idx_mel = idx.clamp(max=self.T + self.lfr_half - 1)
        self.idx_audio = torch.arange(NFFT).unsqueeze(0) + torch.arange(0, self.L - NFFT + 1, HOP).unsqueeze(-1)
        self.inv_ref_air = float(1.0 / (math.sqrt(self.L) * 2e-5))


def encoder(w, x, caches):
    """FunASR FSMN encoder with explicit caches: x [B,T,400], caches 4 x [B,128,19,1]
    -> (p_silence [B,T], new caches).  ref: FSMN/modeling_modified/encoder.py:208-217 (FSMN.forward),
    :139-144 (FsmnStack), :108-110 (BasicBlock), :78-83 (FSMNBlock)."""
    h = F.linear(x, w["in1_w"], w["in1_b"])
    h = F.relu(F.linear(h, w["in2_w"], w["in2_b"]))
    new_caches = []
    for l in range(4):
        p = F.linear(h, w[f"l{l}_lin_w"])                                   # [B,T,128]
        seq = torch.cat((caches[l].squeeze(-1), p.transpose(1, 2)), dim=2)  # [B,128,19+T]
        new_caches.append(seq[:, :, -(LORDER - 1):].unsqueeze(-1))
        fir = F.conv1d(seq, w[f"l{l}_fir_w"].unsqueeze(1), groups=PROJ)     # [B,128,T]
        m = p + fir.transpose(1, 2)
        h = F.relu(F.linear(m, w[f"l{l}_aff_w"], w[f"l{l}_aff_b"]))
    o = F.linear(F.linear(h, w["out1_w"], w["out1_b"]), w["out2_w"], w["out2_b"])
    return torch.softmax(o, dim=-1)[..., 0], new_caches


def features(fe, audio_i16):
    """int16 [B,1,L] -> (prepped audio [B,1,L], LFR+CMVN-ready log-mel [B,T,400] before CMVN).
    ref: FSMN/Export_FSMN_VAD.py:76-85."""
    a = ostft.prep_fsmn(audio_i16)
    re, im = ostft.stft(a, fe.cos_k, fe.sin_k, HOP, True, "constant")
    m = omel.log_mel(re, im, fe.fbank, 1e-5, "clamp").transpose(1, 2)       # [B,T,80]
    left = m[:, :1, :].expand(-1, fe.lfr_half, -1)
    padded = torch.cat((left, m), dim=1)
    lfr = padded[:, fe.idx_mel].reshape(m.shape[0], fe.T_lfr, -1)
    return a, lfr


def forward(fe, w, audio_i16, caches, one_minus_speech_threshold, noise_average_dB,
            speech_2_noise_ratio=1.0, return_raw=False):
    """One session.run equivalent, batched: returns (score uint8 [B,T], caches, noisy_dB [B]).
    ref: FSMN/Export_FSMN_VAD.py:75-101."""
    a, lfr = features(fe, audio_i16)
    p_sil, caches = encoder(w, (lfr + w["cmvn_means"]) * w["cmvn_vars"], caches)
    score = p_sil
    if speech_2_noise_ratio > 1.0:
        score = score + torch.pow(score, speech_2_noise_ratio)
    elif speech_2_noise_ratio < 1.0:
        score = score + 1.0
    else:
        score = score + score
    frames = (a * fe.inv_ref_air).squeeze(1)[:, fe.idx_audio]               # [B,97,512]
    power_dB = torch.log10(torch.sum(frames * frames, dim=-1) + 0.00002)
    T = score.shape[-1]
    power_dB = torch.cat((power_dB, power_dB[:, -1:].expand(-1, T - power_dB.shape[-1])), dim=-1)
    thr = torch.as_tensor(one_minus_speech_threshold, dtype=torch.float32).reshape(-1, 1)
    nz = torch.as_tensor(noise_average_dB, dtype=torch.float32).reshape(-1, 1)
    cond = (score <= thr) & (power_dB >= nz)
    noisy = torch.stack([power_dB[b][~cond[b]].mean() for b in range(cond.shape[0])])
    if return_raw:
        return cond.to(torch.uint8), caches, noisy, score, power_dB
    return cond.to(torch.uint8), caches, noisy


def run_clip(fe, w, audio_i16_1d, pad_noise, *, look_backward_s=0.3, speaking=0.5, silence_score=0.5,
             snr_threshold=10.0, noise_init_dB=30.0, one_minus_speech_threshold=1.0,
             fusion=0.3, min_speech=0.2):
    """Whole-clip driver for ONE clip: window grid, per-chunk forward with cache + noise-floor
    feedback, look-ahead vote, tail, timestamps.
    ref: FSMN/Inference_FSMN_VAD_ONNX.py:68-99 (prep), :156-234 (loop), :239-240 (timestamps).
    `audio_i16_1d` is already peak-normalised int16; `pad_noise` replaces the unseeded RNG."""
    L = fe.L
    frame = 160
    lb = int(look_backward_s * SR // frame)
    stride = L - (lb + 1) * frame
    slide = fe.T - lb
    audio, _ = postproc.pad_to_window_grid(audio_i16_1d, L, stride, pad_noise)
    aligned = audio.shape[0]
    caches = [torch.zeros(1, PROJ, LORDER - 1, 1) for _ in range(4)]
    noise_dB = np.array([noise_init_dB + snr_threshold], dtype=np.float32) * np.float32(0.1)
    snr = snr_threshold * 0.1
    thr = np.array([one_minus_speech_threshold], dtype=np.float32)
    silence = True
    saved = []
    s = 0
    score = None
    while s + L <= aligned:
        chunk = torch.from_numpy(audio[s:s + L].copy()).reshape(1, 1, -1)
        sc, caches, noisy = forward(fe, w, chunk, caches, thr, noise_dB)
        score = sc[0].numpy()
        flags, silence = postproc.lookahead_vote(score, slide, lb if lb else 1, speaking, silence_score, silence)
        saved += flags
        nd = noisy.numpy()[0]
        if nd > 0.0:
            noise_dB = 0.5 * (noise_dB + nd + snr)
        s += stride
    flags, silence = postproc.tail_flags_fsmn(score, slide, fe.T, silence)
    saved += flags
    ts = postproc.vad_to_timestamps(saved, frame / SR)
    return postproc.process_timestamps(ts, fusion, min_speech), saved
