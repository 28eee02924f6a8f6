"""Audio ingest with the arithmetic of the reference drivers' pydub call chain
`AudioSegment.from_file(p).set_channels(1).set_frame_rate(16000).get_array_of_samples()`
(e.g. FSMN/Inference_FSMN_VAD_ONNX.py:68): pydub delegates to the stdlib `audioop`
(`tomono` with 0.5/0.5, then `ratecv`), which is what this does for PCM wav files.

`audioop` left the standard library in Python 3.13: when it cannot be imported the same integer arithmetic runs in numpy
(`_tomono`, `_ratecv`: the closed form of ratecv's phase walk that csrc/ingest.hip uses on the device; bit-identical to
audioop on 16-bit PCM, tests/test_cabi_cpu.py).  Only PCM wav is read -- the reference's pydub path hands anything
ffmpeg can decode to the same two calls; decode such files to wav first."""
from __future__ import annotations

import math
import wave

import numpy as np

try:
    import audioop as _audioop
except ImportError:              # Python >= 3.13
    _audioop = None


def _lin2lin16(data, width):
    """bytes of `width`-byte little-endian PCM -> int16 numpy (audioop.lin2lin(data, width, 2): keep the top 16 bits;
    8-bit wav is unsigned in the file but audioop treats the bytes as signed, like pydub's chain does after its own bias)."""
    if width == 2:
        return np.frombuffer(data, dtype="<i2").astype(np.int16)
    if width == 1:
        return (np.frombuffer(data, dtype=np.int8).astype(np.int16) << 8).astype(np.int16)
    if width == 4:
        return (np.frombuffer(data, dtype="<i4") >> 16).astype(np.int16)
    if width == 3:
        b = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3)
        return (b[:, 1].astype(np.int16) | (b[:, 2].astype(np.int8).astype(np.int16) << 8)).astype(np.int16)
    raise ValueError(f"unsupported sample width {width}")


def _tomono(x):
    """interleaved stereo int16 -> mono: floor(L * 0.5 + R * 0.5) (audioop.tomono(data, 2, 0.5, 0.5))"""
    v = x.astype(np.int64).reshape(-1, 2)
    return ((v[:, 0] + v[:, 1]) >> 1).astype(np.int16)


def _ratecv(x, in_rate, out_rate):
    """audioop.ratecv(data, 2, 1, in_rate, out_rate, None)[0] in closed form: output e uses m = ceil(e*I/O) + 1 consumed
    frames, phase d = (m-1)*O - e*I, out = floor((x[m-2]*d + x[m-1]*(O-d)) / O) with x[-1] = 0 (rates reduced by their gcd)."""
    g = math.gcd(int(in_rate), int(out_rate))
    I, O = int(in_rate) // g, int(out_rate) // g
    n = x.shape[0]
    if n == 0:
        return x.astype(np.int16)
    n_out = (n - 1) * O // I + 1
    e = np.arange(n_out, dtype=np.int64)
    m = (e * I + O - 1) // O + 1
    d = (m - 1) * O - e * I
    xp = np.concatenate(([0], x.astype(np.int64)))           # xp[k + 1] = x[k], xp[0] = the zero before the first frame
    prev, cur = xp[m - 1], xp[m]
    return np.floor_divide(prev * d + cur * (O - d), O).astype(np.int16)


def load_wav(path, sample_rate=16000, channels=1):
    """-> int16 numpy [n] (mono) at `sample_rate`."""
    with wave.open(path, "rb") as w:
        nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        data = w.readframes(n)
    if _audioop is not None:
        if width != 2:
            data = _audioop.lin2lin(data, width, 2)
        if channels == 1 and nch == 2:
            data = _audioop.tomono(data, 2, 0.5, 0.5)
            nch = 1
        elif nch != channels:
            raise ValueError(f"unsupported channel conversion {nch} -> {channels}")
        if rate != sample_rate:
            data, _ = _audioop.ratecv(data, 2, nch, rate, sample_rate, None)
        return np.frombuffer(data, dtype=np.int16).copy()
    x = _lin2lin16(data, width)
    if channels == 1 and nch == 2:
        x = _tomono(x)
    elif nch != channels:
        raise ValueError(f"unsupported channel conversion {nch} -> {channels}")
    if rate != sample_rate:
        if channels != 1:
            raise ValueError("rate conversion of multi-channel audio needs the stdlib audioop")
        x = _ratecv(x, rate, sample_rate)
    return np.ascontiguousarray(x, dtype=np.int16)


def read_wav_raw(path):
    """-> (int16 numpy [frames * channels] interleaved, channels, rate) without any conversion (16-bit PCM only)."""
    with wave.open(path, "rb") as w:
        nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        data = w.readframes(n)
    return _lin2lin16(data, width).copy(), nch, rate


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
