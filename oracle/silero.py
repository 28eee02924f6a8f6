"""Oracle: Silero path.

TEST INFRASTRUCTURE -- CPU restatement in torch float32.

a12 (the network) lives in the un-vendored pip package `silero_vad` (version unpinned;
file silero_vad/data/silero_vad.onnx, call site Silero/modeling_modified/utils_vad.py:117,
Silero/Export_Silero_VAD.py:91).  It is NOT in /root/reference, so the graph below restates the
PUBLISHED silero-vad v5 16 kHz architecture => **PARITY UNPINNED** for the network itself:

    x[B,576] --reflect-pad right 64--> [B,640]
      --conv1d(basis[258,1,256], stride 128)--> [B,258,4] -> sqrt(re^2+im^2) [B,129,4]
      --Conv1d(129,128,3,s1,p1)+ReLU -> Conv1d(128,64,3,s2,p1)+ReLU
      --Conv1d(64,64,3,s2,p1)+ReLU  -> Conv1d(64,128,3,s1,p1)+ReLU            [B,128,1]
      --LSTMCell(128,128) with state=(h,c)=state[0],state[1]
      --ReLU -> Conv1d(128,1,1) -> sigmoid -> mean over time                  [B,1]

What the reference DOES pin (and this file follows exactly): the boundary -- input [B,576] f32 =
64-sample context + 512 new samples, state [2,B,128], sr int64, outputs (out[B,1], stateN) --
and the wrapper / segmenter logic around it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import postproc

CONTEXT = 64
WINDOW = 512
HIDDEN = 128


def net_forward(w, x, state):
    """One ORT-equivalent call: x [B,576] f32, state [2,B,128] -> (out [B,1], stateN [2,B,128])."""
    xp = F.pad(x.unsqueeze(1), (0, 64), mode="reflect")                 # [B,1,640]
    spec = F.conv1d(xp, w["stft_basis"].unsqueeze(1), stride=128)       # [B,258,4]
    re, im = spec[:, :129], spec[:, 129:]
    y = torch.sqrt(re * re + im * im)                                   # [B,129,4]
    strides = (1, 2, 2, 1)
    for i in range(4):
        y = F.relu(F.conv1d(y, w[f"enc{i}_w"], w[f"enc{i}_b"], stride=strides[i], padding=1))
    feat = y.squeeze(-1)                                                # [B,128]
    h0, c0 = state[0], state[1]
    gates = feat @ w["lstm_w_ih"].t() + w["lstm_b_ih"] + h0 @ w["lstm_w_hh"].t() + w["lstm_b_hh"]
    i_g, f_g, g_g, o_g = gates.chunk(4, dim=1)                          # torch LSTMCell gate order
    c1 = torch.sigmoid(f_g) * c0 + torch.sigmoid(i_g) * torch.tanh(g_g)
    h1 = torch.sigmoid(o_g) * torch.tanh(c1)
    logit = F.relu(h1) @ w["dec_w"].reshape(-1, 1) + w["dec_b"]         # [B,1]
    return torch.sigmoid(logit), torch.stack([h1, c1])


def input_projection(w, x):
    """The state-independent part of one call, in the dtype of `w` / `x` (tests evaluate it in float64 to rank two kernels' float32
    errors): x [B,576] -> W_ih feat + b_ih + b_hh [B,512] in torch gate order, the quantity the encoder kernels hand the recurrent
    kernel.  Same graph as net_forward up to `feat`."""
    xp = F.pad(x.unsqueeze(1), (0, 64), mode="reflect")
    spec = F.conv1d(xp, w["stft_basis"].unsqueeze(1), stride=128)
    re, im = spec[:, :129], spec[:, 129:]
    y = torch.sqrt(re * re + im * im)
    strides = (1, 2, 2, 1)
    for i in range(4):
        y = F.relu(F.conv1d(y, w[f"enc{i}_w"], w[f"enc{i}_b"], stride=strides[i], padding=1))
    return y.squeeze(-1) @ w["lstm_w_ih"].t() + w["lstm_b_ih"] + w["lstm_b_hh"]


class OnnxWrapperOracle:
    """State/context carry around the network call.
    ref: Silero/modeling_modified/utils_vad.py:69-146 (validation :69-85, reset :87-91,
    __call__ :93-128, audio_forward :130-146)."""

    def __init__(self, weights):
        self.w = weights
        self.sample_rates = [16000]
        self.reset_states()

    def _validate_input(self, x, sr):
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if x.dim() > 2:
            raise ValueError(f"Too many dimensions for input audio chunk {x.dim()}")
        if sr != 16000 and (sr % 16000 == 0):
            x = x[:, ::sr // 16000]
            sr = 16000
        if sr not in self.sample_rates:
            raise ValueError(f"Supported sampling rates: {self.sample_rates} (or multiply of 16000)")
        if sr / x.shape[1] > 31.25:
            raise ValueError("Input audio chunk is too short")
        return x, sr

    def reset_states(self, batch_size=1):
        self._state = torch.zeros((2, batch_size, HIDDEN)).float()
        self._context = torch.zeros(0)
        self._last_sr = 0
        self._last_batch_size = 0

    def __call__(self, x, sr):
        x, sr = self._validate_input(x, sr)
        if x.shape[-1] != WINDOW:
            raise ValueError(f"Provided number of samples is {x.shape[-1]}")
        b = x.shape[0]
        if not self._last_batch_size:
            self.reset_states(b)
        if self._last_sr and self._last_sr != sr:
            self.reset_states(b)
        if self._last_batch_size and self._last_batch_size != b:
            self.reset_states(b)
        if not len(self._context):
            self._context = torch.zeros(b, CONTEXT)
        x = torch.cat([self._context, x], dim=1)
        out, self._state = net_forward(self.w, x, self._state)
        self._context = x[..., -CONTEXT:]
        self._last_sr = sr
        self._last_batch_size = b
        return out

    def audio_forward(self, x, sr):
        x, sr = self._validate_input(x, sr)
        self.reset_states()
        if x.shape[1] % WINDOW:
            x = F.pad(x, (0, WINDOW - x.shape[1] % WINDOW), "constant", value=0.0)
        outs = [self(x[:, i:i + WINDOW], sr) for i in range(0, x.shape[1], WINDOW)]
        return torch.cat(outs, dim=1)


@torch.no_grad()
def speech_probs(audio, model, sampling_rate=16000):
    """The model loop of get_speech_timestamps. ref: utils_vad.py:350, 359-372."""
    model.reset_states()
    n = len(audio)
    probs = []
    for s in range(0, n, WINDOW):
        chunk = audio[s:s + WINDOW]
        if len(chunk) < WINDOW:
            chunk = F.pad(chunk, (0, int(WINDOW - len(chunk))))
        probs.append(model(chunk, sampling_rate).item())
    return probs


@torch.no_grad()
def get_speech_timestamps(audio, model, **kw):
    """ref: Silero/modeling_modified/utils_vad.py:248-491 (16 kHz, mono path)."""
    audio = torch.as_tensor(audio)
    while audio.dim() > 1 and audio.shape[0] == 1:
        audio = audio.squeeze(0)
    if audio.dim() > 1:
        raise ValueError("More than one dimension in audio. Are you trying to process audio with 2 channels?")
    sr = kw.get("sampling_rate", 16000)
    probs = speech_probs(audio, model, sr)
    return postproc.silero_segments(probs, len(audio), **kw)
