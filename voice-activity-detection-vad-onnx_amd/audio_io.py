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
