"""Oracle: audio ingest.  TEST INFRASTRUCTURE -- numpy restatement of what the reference drivers'
pydub chain does to 16-bit PCM: `AudioSegment.set_channels(1)` = audioop.tomono(data, 2, 0.5, 0.5) and
`set_frame_rate(r)` = audioop.ratecv(data, 2, 1, in_rate, r, None)[0]  (ref: FSMN/Inference_FSMN_VAD_ONNX.py:68,
Silero/Inference_Silero_VAD_ONNX.py:83 and the other drivers).  pydub / audioop are third-party / stdlib, absent from
/root/reference: pinned against the stdlib `audioop` itself in tests/test_oracle_golden.py."""
from __future__ import annotations

import math

import numpy as np


def tomono(interleaved_i16):
    """[frames*2] int16 -> [frames] int16: floor(L*0.5 + R*0.5) (audioop.tomono)."""
    x = np.asarray(interleaved_i16, dtype=np.int64).reshape(-1, 2)
    return ((x[:, 0] + x[:, 1]) >> 1).astype(np.int16)


def ratecv(mono_i16, in_rate, out_rate):
    """audioop.ratecv's phase walk, frame by frame (state None, weightA = 1, weightB = 0)."""
    x = np.asarray(mono_i16, dtype=np.int64)
    g = math.gcd(in_rate, out_rate)
    I, O = in_rate // g, out_rate // g
    d, prev, cur, out, k = -O, 0, 0, [], 0
    n = len(x)
    while True:
        while d < 0:
            if k == n:
                return np.array(out, dtype=np.int16)
            prev, cur = cur, int(x[k]) << 16
            k += 1
            d += O
        while d >= 0:
            cur_o = int((float(prev) * float(d) + float(cur) * float(O - d)) / float(O))      # C cast: truncation
            out.append(cur_o >> 16)
            d -= I


def ingest(interleaved_i16, channels, in_rate, out_rate=16000):
    x = tomono(interleaved_i16) if channels == 2 else np.asarray(interleaved_i16, dtype=np.int16)
    return ratecv(x, in_rate, out_rate) if in_rate != out_rate else x
