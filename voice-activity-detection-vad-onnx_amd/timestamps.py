"""Host-side timestamp helpers shared by every driver (reference rows a10, a22).

Same names, argument meaning and results as the reference's copies
(FSMN/Inference_FSMN_VAD_ONNX.py:60-63,102-153,244-258 -- duplicated verbatim in 8-12 scripts
there); implemented once here, numpy where it helps.
"""
from __future__ import annotations

from datetime import timedelta

import numpy as np


def normalize_to_int16(audio):
    """Peak-normalise to the int16 range (FSMN/Inference_FSMN_VAD_ONNX.py:60-63)."""
    peak = np.max(np.abs(audio))
    scale = 32767.0 / peak if peak > 0 else 1.0
    return (audio * float(scale)).astype(np.int16)


def normalise_audio(audio, target_rms=8192.0):
    """Optional RMS normalisation (Inference_NVIDIA_MarbleNet_VAD_ONNX.py:110-118)."""
    buf = audio.astype(np.float32)
    rms = np.sqrt(np.mean(buf * buf, dtype=np.float32), dtype=np.float32)
    if not rms > 0:
        return audio
    buf *= (target_rms / (rms + 1e-7))
    np.clip(buf, -32768.0, 32767.0, out=buf)
    return buf.astype(np.int16)


def vad_to_timestamps(vad_output, frame_duration):
    """Silence flags -> (start, end) seconds; start = i*fd, end = i*fd + fd."""
    flags = np.asarray(vad_output, dtype=bool)
    n = flags.shape[0]
    if n == 0:
        return []
    speech = np.concatenate(([False], ~flags, [False]))
    edges = np.flatnonzero(speech[1:] != speech[:-1])
    out = []
    for a, b in zip(edges[0::2].tolist(), edges[1::2].tolist()):
        # b is the first silent frame after the run (or n when speech runs to the end)
        out.append((a * frame_duration, n * frame_duration if b == n else b * frame_duration + frame_duration))
    return out


def process_timestamps(timestamps, fusion_threshold=1.0, min_duration=0.5):
    """Drop segments shorter than min_duration, then merge neighbours closer than
    fusion_threshold -- two passes, as the reference does."""
    segs = [(s, e) for s, e in timestamps if (e - s) >= min_duration]
    for _ in range(2):
        merged = []
        for s, e in segs:
            if merged and (s - merged[-1][1] <= fusion_threshold):
                merged[-1] = (merged[-1][0], e)
            else:
                merged.append((s, e))
        segs = merged
    return segs


def format_time(seconds):
    """'hh:mm:ss.mmm' (milliseconds truncated after timedelta's microsecond rounding)."""
    t = timedelta(seconds=seconds).total_seconds()
    whole = int(t)
    ms = int((t - whole) * 1000)
    return "%02d:%02d:%02d.%03d" % (whole // 3600, (whole % 3600) // 60, whole % 60, ms)


def write_timestamp_files(timestamps, sample_rate, path_second, path_indices, echo=print):
    """The two text files every reference driver writes."""
    with open(path_second, "w", encoding="UTF-8") as fh:
        echo("\nTimestamps in Second:")
        for s, e in timestamps:
            line = f"{format_time(s)} --> {format_time(e)}\n"
            fh.write(line)
            echo(line.rstrip("\n"))
    with open(path_indices, "w", encoding="UTF-8") as fh:
        echo("\nTimestamps in Indices:")
        for s, e in timestamps:
            line = f"{int(s * sample_rate)} --> {int(e * sample_rate)}\n"
            fh.write(line)
            echo(line.rstrip("\n"))


def indices(timestamps, sample_rate):
    """Integer sample-index pairs exactly as written to timestamps_indices.txt."""
    return [(int(s * sample_rate), int(e * sample_rate)) for s, e in timestamps]
