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


def log_mel(fe, audio_i16, in_sample_rate=16000):
    """int16 [B,1,L] -> [B,80,T], T = (L'-400)//160+1 with L' the window length after the in-graph resample.
    ref: Export_FireRedVAD.py:428-461."""
    a = ostft.prep_two_tap(audio_i16, 1.0, in_sample_rate)
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


def forward(fe, w, audio_i16, in_sample_rate=16000):
    """session.run equivalent: int16 [B,1,16000] -> probs [B,odim,98]. ref: Export_FireRedVAD.py:420-467."""
    return detect_model(w, log_mel(fe, audio_i16, in_sample_rate))


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


def run_clip_aed(fe, w, audio_i16_1d, pad_noise, window=16000, thresholds=(0.4, 0.5, 0.5), post=(5, 20, 2000, 20, 5, 0)):
    """AED driver for ONE clip (odim == 3): per-event VadPostprocessor + ratio of frames over threshold.
    ref: FireRedVAD/Inference_FireRed_ONNX.py:620-742."""
    n = int(np.asarray(audio_i16_1d).shape[0])
    audio, _ = postproc.pad_to_window_grid(audio_i16_1d, window, window, pad_noise)
    probs = []
    for s in range(0, audio.shape[0] - window + 1, window):
        chunk = torch.from_numpy(audio[s:s + window].copy()).reshape(1, 1, -1)
        probs.append(forward(fe, w, chunk)[0].numpy())
    allp = np.concatenate(probs, axis=1)[:, :valid_frame_count(n)] if probs else np.zeros((3, 0), np.float32)
    ts, ratio = {}, {}
    for idx, event in enumerate(("speech", "singing", "music")):
        pp = postproc.VadPostprocessor(post[0], thresholds[idx], *post[1:])
        ts[event] = pp.decision_to_segment(pp.process(allp[idx]), n / SR)
        ratio[event] = round(float(np.mean(allp[idx] >= thresholds[idx])) if allp.shape[1] else 0.0, 3)
    return ts, ratio, allp


# ---------------------------------------------------------------------------- streaming variant (§8f-1)
def detect_model_stream(w, feat, caches_in):
    """Cache-carrying DFSMN (no look-ahead). feat [1,80,T], caches [R,1,P,(N1-1)*S1] -> (probs [1,odim,T], caches_out).
    ref: FireRedVAD/Export_FireRedVAD.py:479-612."""
    c = w["cfg"]
    N1, S1 = c["N1"], c["S1"]
    pad = (N1 - 1) * S1

    def pw(x, wk, bk=None):
        return F.conv1d(x, w[wk].unsqueeze(-1), None if bk is None else w[bk])

    def mem_stream(x, lb_w, cache):
        seq = torch.cat([cache, x], dim=2)
        return x + F.conv1d(seq, lb_w.unsqueeze(1), dilation=S1, groups=x.shape[1]), seq[:, :, -pad:]

    h = F.relu(pw(feat, "fc1_w", "fc1_b"))
    p = F.relu(pw(h, "fc2_w", "fc2_b"))
    mem, c0 = mem_stream(p, w["fsmn0_lb"], caches_in[0])
    new = [c0]
    for r in range(1, c["R"]):
        pp = pw(F.relu(pw(mem, f"blk{r}_fc1_w", f"blk{r}_fc1_b")), f"blk{r}_fc2_w")
        m2, cr = mem_stream(pp, w[f"fsmn{r}_lb"], caches_in[r])
        mem = m2 + mem
        new.append(cr)
    x = mem
    for m in range(c["M"]):
        x = F.relu(pw(x, f"dnn{m}_w", f"dnn{m}_b"))
    return torch.sigmoid(pw(x, "out_w", "out_b")), torch.stack(new, dim=0)


def forward_stream(fe, w, audio_i16, caches_in):
    """ref: FireRedStreamVAD_ONNX.forward, Export_FireRedVAD.py:700-749."""
    return detect_model_stream(w, log_mel(fe, audio_i16), caches_in)


class StreamVadPostprocessor:
    """Frame-by-frame streaming decision logic with a circular smoothing buffer.
    ref: FireRedVAD/Export_FireRedVAD.py:1161-1454 (process_batch :1205-1339)."""

    def __init__(self, smooth_window_size, speech_threshold, pad_start_frame, min_speech_frame,
                 max_speech_frame, min_silence_frame, frames_per_second=100):
        self.ws = max(1, smooth_window_size)
        self.thr = np.float32(speech_threshold)
        self.pad_start = max(self.ws, pad_start_frame)
        self.min_sp, self.max_sp, self.min_si = min_speech_frame, max_speech_frame, min_silence_frame
        self.fps = frames_per_second
        self.reset()

    def reset(self):
        self.buf = np.zeros(self.ws, dtype=np.float32)
        self.buf_sum = np.float32(0.0)
        self.pos = self.count = self.frame_cnt = 0
        self.state = 0
        self.speech_cnt = self.silence_cnt = 0
        self.hit_max = False
        self.last_start = self.last_end = -1

    def process_batch(self, raw_probs):
        probs = np.asarray(raw_probs, dtype=np.float32) if not isinstance(raw_probs, np.ndarray) else raw_probs
        if probs.shape[0] == 0:
            return []
        inv = 1.0 / self.fps
        out = []
        for p in probs:
            self.frame_cnt += 1
            fc = self.frame_cnt
            if self.ws <= 1:
                sm = p
            else:
                old = self.buf[self.pos]
                self.buf[self.pos] = p
                self.buf_sum += p - old
                self.pos = (self.pos + 1) % self.ws
                if self.count < self.ws:
                    self.count += 1
                sm = self.buf_sum / self.count
            hot = 1 if sm >= self.thr else 0
            s_out = e_out = -1
            if self.hit_max:
                s_out = fc
                self.last_start = fc
                self.hit_max = False
            if self.state == 0:
                if hot:
                    self.state, self.speech_cnt = 1, 1
                else:
                    self.silence_cnt += 1
                    self.speech_cnt = 0
            elif self.state == 1:
                if hot:
                    self.speech_cnt += 1
                    if self.speech_cnt >= self.min_sp:
                        self.state = 2
                        s_out = max(1, fc - self.speech_cnt + 1 - self.pad_start, self.last_end + 1)
                        self.last_start = s_out
                        self.silence_cnt = 0
                else:
                    self.state, self.silence_cnt, self.speech_cnt = 0, 1, 0
            elif self.state == 2:
                self.speech_cnt += 1
                if hot:
                    self.silence_cnt = 0
                    if self.speech_cnt >= self.max_sp:
                        self.hit_max, self.speech_cnt = True, 0
                        e_out, s_out = fc, self.last_start
                        self.last_start, self.last_end = -1, fc
                else:
                    self.state, self.silence_cnt = 3, 1
            else:
                self.speech_cnt += 1
                if hot:
                    self.state, self.silence_cnt = 2, 0
                    if self.speech_cnt >= self.max_sp:
                        self.hit_max, self.speech_cnt = True, 0
                        e_out, s_out = fc, self.last_start
                        self.last_start, self.last_end = -1, fc
                else:
                    self.silence_cnt += 1
                    if self.silence_cnt >= self.min_si:
                        self.state = 0
                        e_out, s_out = fc, self.last_start
                        self.last_end, self.last_start = fc, -1
                        self.speech_cnt = 0
            if e_out > 0 and s_out > 0:
                out.append((max(0, s_out - 1) * inv, max(0, e_out - 1) * inv))
        if self.last_start > 0:
            out.append((max(0, self.last_start - 1) * inv, (self.frame_cnt - 1) * inv))
        return out


def run_clip_stream(fe, w, audio_i16_1d, chunk=2560, post=(5, 0.4, 5, 8, 2000, 20)):
    """ref: FireRedVAD/Inference_FireRed_ONNX.py:767-822."""
    a = np.asarray(audio_i16_1d)
    n = a.shape[0]
    c = w["cfg"]
    caches = torch.zeros(c["R"], 1, c["P"], (c["N1"] - 1) * c["S1"])
    probs, pos = [], 0
    while pos < n:
        end = min(pos + chunk, n)
        x = a[pos:end]
        if len(x) < WIN:
            x = np.pad(x, (0, WIN - len(x)), mode="constant")
        pr, caches = forward_stream(fe, w, torch.from_numpy(x.copy()).reshape(1, 1, -1), caches)
        probs.append(pr[0, 0].numpy())
        pos = end
    allp = np.concatenate(probs, axis=0)[:valid_frame_count(n)] if probs else np.zeros((0,), np.float32)
    return StreamVadPostprocessor(*post).process_batch(allp), allp
