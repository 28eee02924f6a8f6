"""Audio ingest with the arithmetic of the reference drivers' pydub call chain
`AudioSegment.from_file(p).set_channels(1).set_frame_rate(16000).get_array_of_samples()`
(e.g. FSMN/Inference_FSMN_VAD_ONNX.py:68): pydub delegates to the stdlib `audioop`
(`tomono` with 0.5/0.5, then `ratecv`), which is what this does for PCM wav files."""
from __future__ import annotations

import audioop
import wave

import numpy as np


def load_wav(path, sample_rate=16000, channels=1):
    """-> int16 numpy [n] (mono) at `sample_rate`."""
    with wave.open(path, "rb") as w:
        nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        data = w.readframes(n)
    if width != 2:
        data = audioop.lin2lin(data, width, 2)
        width = 2
    if channels == 1 and nch == 2:
        data = audioop.tomono(data, width, 0.5, 0.5)
        nch = 1
    elif nch != channels:
        raise ValueError(f"unsupported channel conversion {nch} -> {channels}")
    if rate != sample_rate:
        data, _ = audioop.ratecv(data, width, nch, rate, sample_rate, None)
    return np.frombuffer(data, dtype=np.int16).copy()


def read_wav_raw(path):
    """-> (int16 numpy [frames * channels] interleaved, channels, rate) without any conversion (16-bit PCM only)."""
    with wave.open(path, "rb") as w:
        nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        data = w.readframes(n)
    if width != 2:
        data = audioop.lin2lin(data, width, 2)
    return np.frombuffer(data, dtype=np.int16).copy(), nch, rate


def ingest_device(raw_i16, channels, rate, sample_rate=16000, device="cuda:0"):
    """Batch of equal-length raw clips int16 [B, frames * channels] (interleaved) -> device int16 [B, n] mono at
    `sample_rate`, with audioop's tomono + ratecv arithmetic on the GPU (vadx_ingest_pcm16): the upload is the raw
    file bytes and the down-mix / resampling never touches the host."""
    from . import _lib
    t = _lib.require_gpu()
    x = raw_i16 if t.is_tensor(raw_i16) else t.from_numpy(np.ascontiguousarray(raw_i16, dtype=np.int16))
    if x.dim() == 1:
        x = x.unsqueeze(0)
    x = x.to(device).contiguous()
    B, n = x.shape
    if n % channels:
        raise ValueError("row length is not a whole number of frames")
    frames = n // channels
    L = _lib.lib()
    nout = L.vadx_ingest_out_frames(frames, int(rate), int(sample_rate))
    out = t.empty((B, nout), dtype=t.int16, device=x.device)
    with t.cuda.device(x.device):
        _lib.check(L.vadx_ingest_pcm16(x.data_ptr(), _lib.row_stride(x), int(channels), frames, int(rate), int(sample_rate),
                                       out.data_ptr(), _lib.row_stride(out), B, _lib.stream_ptr()))
    return out
