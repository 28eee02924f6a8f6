"""Oracle: host-side decision / timestamp logic.

TEST INFRASTRUCTURE -- CPU restatement, never imported by the product package.
Plain Python / numpy; semantics follow the reference line by line, code is our own.
"""
from __future__ import annotations

from datetime import timedelta

import numpy as np


# --------------------------------------------------------------------------- a10
def vad_to_timestamps(silence_flags, frame_duration):
    """bool[] (True = silence) -> [(start_s, end_s)].
    ref: FSMN/Inference_FSMN_VAD_ONNX.py:125-141 (8 identical copies)."""
    out = []
    seg_start = None
    n = 0
    for i, is_sil in enumerate(silence_flags):
        n = i + 1
        if is_sil:
            if seg_start is not None:
                out.append((seg_start, i * frame_duration + frame_duration))
                seg_start = None
        elif seg_start is None:
            seg_start = i * frame_duration
    if seg_start is not None:
        out.append((seg_start, n * frame_duration))
    return out


def _fuse(segs, gap):
    fused = []
    for s, e in segs:
        if fused and (s - fused[-1][1] <= gap):
            fused[-1] = (fused[-1][0], e)
        else:
            fused.append((s, e))
    return fused


def process_timestamps(timestamps, fusion_threshold=1.0, min_duration=0.5):
    """Drop short segments, then fuse near neighbours twice.
    ref: FSMN/Inference_FSMN_VAD_ONNX.py:102-122."""
    kept = [(s, e) for s, e in timestamps if (e - s) >= min_duration]
    return _fuse(_fuse(kept, fusion_threshold), fusion_threshold)


def format_time(seconds):
    """'hh:mm:ss.mmm' with millisecond truncation through timedelta's microsecond rounding.
    ref: FSMN/Inference_FSMN_VAD_ONNX.py:144-153."""
    total = timedelta(seconds=seconds).total_seconds()
    whole = int(total)
    ms = int((total - whole) * 1000)
    return f"{whole // 3600:02}:{(whole % 3600) // 60:02}:{whole % 60:02}.{ms:03}"


def timestamp_lines(timestamps, sample_rate):
    """The two text files every driver writes. ref: FSMN/Inference_FSMN_VAD_ONNX.py:244-258."""
    sec = [f"{format_time(s)} --> {format_time(e)}\n" for s, e in timestamps]
    idx = [f"{int(s * sample_rate)} --> {int(e * sample_rate)}\n" for s, e in timestamps]
    return sec, idx


# --------------------------------------------------------------------------- a22
def normalize_to_int16(audio):
    """Peak-normalise to +-32767. ref: FSMN/Inference_FSMN_VAD_ONNX.py:60-63."""
    peak = np.max(np.abs(audio))
    k = 32767.0 / peak if peak > 0 else 1.0
    return (audio * float(k)).astype(np.int16)


def normalise_audio(audio, target_rms=8192.0):
    """Optional RMS normalisation. ref: Inference_NVIDIA_MarbleNet_VAD_ONNX.py:110-118."""
    x = audio.astype(np.float32)
    rms = np.sqrt(np.mean(x * x, dtype=np.float32), dtype=np.float32)
    if rms > 0:
        x *= (target_rms / (rms + 1e-7))
        np.clip(x, -32768.0, 32767.0, out=x)
        return x.astype(np.int16)
    return audio


def pad_to_window_grid(audio_i16, window, stride, noise):
    """Window-grid alignment with an explicit noise vector instead of the reference's
    unseeded np.random.normal (ref: FSMN/Inference_FSMN_VAD_ONNX.py:88-99).

    `noise` is a float64 standard-normal vector at least as long as the pad; the reference
    scales it by the RMS of the last `pad` samples (long clip) or of the whole clip (short
    clip) and casts to the audio dtype.
    Returns (padded int16 [N'], pad_amount)."""
    a = np.asarray(audio_i16).reshape(-1)
    n = a.shape[0]
    if n > window:
        num_windows = int(np.ceil((n - window) / stride)) + 1
        pad = (num_windows - 1) * stride + window - n
        if pad == 0:
            # reference: audio[:, :, -0:] is the WHOLE clip and a zero-length noise vector
            return a.copy(), 0
        tail = a[-pad:].astype(np.float32)
        rms = np.sqrt(np.mean(tail * tail))
    elif n < window:
        pad = window - n
        f = a.astype(np.float32)
        rms = np.sqrt(np.mean(f * f))
    else:
        return a.copy(), 0
    fill = (rms * np.asarray(noise[:pad], dtype=np.float64)).astype(a.dtype)
    return np.concatenate((a, fill)), pad


# --------------------------------------------------------------------------- a9
def lookahead_vote(score, slide_range, look_backward, speaking_score, silence_score,
                   silence, active_value=1, inactive_value=0, thresholds=None):
    """One chunk of the FSMN / DFSMN look-ahead majority vote.

    FSMN (uint8 score, ref: FSMN/Inference_FSMN_VAD_ONNX.py:188-215): a frame "votes active"
    when score != 0 while silent, and "votes inactive"... precisely:
        silent : score[i] != 0 -> count j in [1,lb) with score[i+j] != 0, (1+cnt)/lb >= SPEAKING -> speech
        speech : score[i] != 1 -> count j with score[i+j] != 1, (1+cnt)/lb <= SILENCE -> stay speech
    DFSMN (f32 score, ref: DFSMN/near_and_far_end_audio/Inference_DFSMN_VAD_ONNX.py:231-258) uses
    the same loop with predicates score >= 0.5 / score <= 0.5 (pass thresholds=(hi, lo)).
    Returns (list of silence flags, final silence state)."""
    lb = look_backward
    inv_lb = float(1.0 / lb)
    flags = []
    if thresholds is None:
        on = lambda v: v != inactive_value      # noqa: E731  "frame looks active"
        off = lambda v: v != active_value       # noqa: E731  "frame looks inactive"
    else:
        # float32 score against a Python-float constant: NumPy 2 (NEP 50) compares in float32 -- the fixtures were produced by
        # the reference's loop under NumPy 2.2, and a score sitting exactly on float32(0.7) counts as >= 0.7 there
        hi, lo = np.float32(thresholds[0]), np.float32(thresholds[1])
        on = lambda v: np.float32(v) >= hi      # noqa: E731
        off = lambda v: np.float32(v) <= lo     # noqa: E731
    for i in range(slide_range):
        if silence:
            if on(score[i]):
                votes = 1
                for j in range(1, lb):
                    if on(score[i + j]):
                        votes += 1
                silence = not (votes * inv_lb >= speaking_score)
            else:
                silence = True
        else:
            if off(score[i]):
                votes = 1
                for j in range(1, lb):
                    if off(score[i + j]):
                        votes += 1
                silence = not (votes * inv_lb <= silence_score)
            else:
                silence = False
        flags.append(silence)
    return flags, silence


def tail_flags_fsmn(score, start, stop, silence):
    """Plain (no look-ahead) rule for the last frames of the final chunk.
    ref: FSMN/Inference_FSMN_VAD_ONNX.py:223-234."""
    flags = []
    for i in range(start, stop):
        if silence:
            silence = not (score[i] != 0)
        else:
            silence = bool(score[i] != 1)
        flags.append(silence)
    return flags, silence


# --------------------------------------------------------------------------- a13
def silero_segments(speech_probs, audio_length_samples, threshold=0.5, sampling_rate=16000,
                    min_speech_duration_ms=250, max_speech_duration_s=float("inf"),
                    min_silence_duration_ms=100, speech_pad_ms=30, return_seconds=False,
                    time_resolution=1, neg_threshold=None, min_silence_at_max_speech=98,
                    use_max_poss_sil_at_max_speech=True):
    """The segmenter half of get_speech_timestamps: per-window probabilities -> speech dicts.
    ref: Silero/modeling_modified/utils_vad.py:351-482 (the model loop :359-372 lives in
    oracle.silero.get_speech_timestamps)."""
    W = 512 if sampling_rate == 16000 else 256
    min_speech = sampling_rate * min_speech_duration_ms / 1000
    pad = sampling_rate * speech_pad_ms / 1000
    max_speech = sampling_rate * max_speech_duration_s - W - 2 * pad
    min_sil = sampling_rate * min_silence_duration_ms / 1000
    min_sil_at_max = sampling_rate * min_silence_at_max_speech / 1000
    if neg_threshold is None:
        neg_threshold = max(threshold - 0.15, 0.01)

    triggered = False
    speeches = []
    cur = {}
    temp_end = 0
    prev_end = next_start = 0
    possible_ends = []

    for i, p in enumerate(speech_probs):
        pos = W * i
        if (p >= threshold) and temp_end:
            gap = pos - temp_end
            if gap > min_sil_at_max:
                possible_ends.append((temp_end, gap))
            temp_end = 0
            if next_start < prev_end:
                next_start = pos

        if (p >= threshold) and not triggered:
            triggered = True
            cur["start"] = pos
            continue

        if triggered and (pos - cur["start"] > max_speech):
            if use_max_poss_sil_at_max_speech and possible_ends:
                prev_end, dur = max(possible_ends, key=lambda x: x[1])
                cur["end"] = prev_end
                speeches.append(cur)
                cur = {}
                next_start = prev_end + dur
                if next_start < prev_end + pos:
                    cur["start"] = next_start
                else:
                    triggered = False
                prev_end = next_start = temp_end = 0
                possible_ends = []
            else:
                if prev_end:
                    cur["end"] = prev_end
                    speeches.append(cur)
                    cur = {}
                    if next_start < prev_end:
                        triggered = False
                    else:
                        cur["start"] = next_start
                    prev_end = next_start = temp_end = 0
                    possible_ends = []
                else:
                    cur["end"] = pos
                    speeches.append(cur)
                    cur = {}
                    prev_end = next_start = temp_end = 0
                    triggered = False
                    possible_ends = []
                    continue

        if (p < neg_threshold) and triggered:
            if not temp_end:
                temp_end = pos
            sil_now = pos - temp_end
            if (not use_max_poss_sil_at_max_speech) and sil_now > min_sil_at_max:
                prev_end = temp_end
            if sil_now < min_sil:
                continue
            cur["end"] = temp_end
            if (cur["end"] - cur["start"]) > min_speech:
                speeches.append(cur)
            cur = {}
            prev_end = next_start = temp_end = 0
            triggered = False
            possible_ends = []
            continue

    if cur and (audio_length_samples - cur["start"]) > min_speech:
        cur["end"] = audio_length_samples
        speeches.append(cur)

    last = len(speeches) - 1
    for i, sp in enumerate(speeches):
        if i == 0:
            sp["start"] = int(max(0, sp["start"] - pad))
        if i != last:
            gap = speeches[i + 1]["start"] - sp["end"]
            if gap < 2 * pad:
                sp["end"] += int(gap // 2)
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - gap // 2))
            else:
                sp["end"] = int(min(audio_length_samples, sp["end"] + pad))
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - pad))
        else:
            sp["end"] = int(min(audio_length_samples, sp["end"] + pad))

    if return_seconds:
        dur_s = audio_length_samples / sampling_rate
        for sp in speeches:
            sp["start"] = max(round(sp["start"] / sampling_rate, time_resolution), 0)
            sp["end"] = min(round(sp["end"] / sampling_rate, time_resolution), dur_s)
    return speeches


# --------------------------------------------------------------------------- a16
_SIL, _MAYBE_SPEECH, _SPEECH, _MAYBE_SIL = 0, 1, 2, 3


class VadPostprocessor:
    """Smoothing + 4-state machine + start fix + gap merge + dilation + long-segment split.

    ref: FireRedVAD/Inference_FireRed_ONNX.py:102-304 (frame_shift 0.01 s, +0.025 s at end of
    audio) and NVIDIA_.../Inference_NVIDIA_MarbleNet_VAD_ONNX.py:160-353 (takes frame_shift_s, no
    frame-length term).  `frame_length_s=None` selects the MarbleNet flavour."""

    def __init__(self, smooth_window_size, prob_threshold, min_speech_frame, max_speech_frame,
                 min_silence_frame, merge_silence_frame, extend_speech_frame,
                 frame_shift_s=0.01, frame_length_s=0.025):
        self.ws = max(1, smooth_window_size)
        self.thr = np.float32(prob_threshold)
        self.min_speech = min_speech_frame
        self.max_speech = max_speech_frame
        self.min_silence = min_silence_frame
        self.merge = merge_silence_frame
        self.extend = extend_speech_frame
        self.inv_ws = np.float32(1.0 / self.ws)
        self.half_max = max_speech_frame >> 1
        self.shift = np.float32(frame_shift_s)
        self.flen = None if frame_length_s is None else np.float32(frame_length_s)

    # -- smoothing: float32 cumulative sum, expanding mean on the first ws-1 frames
    def smooth(self, probs):
        n = probs.shape[0]
        if self.ws <= 1:
            return probs
        cs = np.empty(n + 1, dtype=np.float32)
        cs[0] = 0.0
        np.cumsum(probs, out=cs[1:])
        sm = np.empty(n, dtype=np.float32)
        for i in range(min(self.ws - 1, n)):
            sm[i] = cs[i + 1] / (i + 1)
        if n >= self.ws:
            sm[self.ws - 1:] = (cs[self.ws:] - cs[:n - self.ws + 1]) * self.inv_ws
        return sm

    def state_machine(self, sm):
        n = sm.shape[0]
        dec = np.zeros(n, dtype=np.int8)
        if self.min_speech <= 0 and self.min_silence <= 0:
            dec[:] = sm >= self.thr
            return dec
        state = _SIL
        t0 = 0
        s0 = 0
        for t in range(n):
            hot = sm[t] >= self.thr
            if state == _SIL:
                if hot:
                    state, t0 = _MAYBE_SPEECH, t
            elif state == _MAYBE_SPEECH:
                if hot:
                    if t - t0 >= self.min_speech:
                        state = _SPEECH
                        dec[t0:t] = 1
                else:
                    state = _SIL
            elif state == _SPEECH:
                if not hot:
                    state, s0 = _MAYBE_SIL, t
            else:
                if not hot:
                    if t - s0 >= self.min_silence:
                        state = _SIL
                else:
                    state = _SPEECH
            dec[t] = 1 if state >= _SPEECH else 0
        return dec

    def fix_starts(self, dec):
        if self.ws <= 1:
            return
        for t in range(1, dec.shape[0]):
            if dec[t] == 1 and dec[t - 1] == 0:
                dec[max(t - self.ws, 0):t] = 1

    def merge_silence(self, dec):
        gap0 = -1
        for t in range(1, dec.shape[0]):
            a, b = dec[t - 1], dec[t]
            if a == 1 and b == 0 and gap0 < 0:
                gap0 = t
            elif a == 0 and b == 1 and gap0 >= 0:
                if t - gap0 < self.merge:
                    dec[gap0:t] = 1
                gap0 = -1

    def dilate(self, dec):
        n = dec.shape[0]
        for order in (range(n), range(n - 1, -1, -1)):
            dist = self.extend + 1
            for t in order:
                if dec[t]:
                    dist = 0
                else:
                    dist += 1
                    if dist <= self.extend:
                        dec[t] = 1

    def split_long(self, dec, probs):
        n = dec.shape[0]
        t = 0
        while t < n:
            if not dec[t]:
                t += 1
                continue
            a = t
            while t < n and dec[t]:
                t += 1
            if t - a > self.max_speech:
                pos, b = a, t
                while pos + self.max_speech < b:
                    lo = pos + self.half_max
                    hi = min(pos + self.max_speech, b)
                    if lo >= hi:
                        break
                    cut = lo + int(np.argmin(probs[lo:hi]))
                    dec[cut] = 0
                    pos = cut + 1

    def process(self, raw_probs):
        probs = np.asarray(raw_probs, dtype=np.float32)
        if probs.shape[0] == 0:
            return np.empty(0, dtype=np.int8)
        dec = self.state_machine(self.smooth(probs))
        self.fix_starts(dec)
        if self.merge > 0:
            self.merge_silence(dec)
        if self.extend > 0:
            self.dilate(dec)
        self.split_long(dec, probs)
        return dec

    def decision_to_segment(self, decisions, wav_dur=None):
        dec = np.asarray(decisions, dtype=np.int8)
        n = dec.shape[0]
        if n == 0:
            return []
        edge = np.diff(np.concatenate(([0], dec, [0])).astype(np.int8))
        starts = np.flatnonzero(edge == 1).astype(np.float32)
        ends = np.flatnonzero(edge == -1).astype(np.float32)
        if starts.shape[0] == 0:
            return []
        seg = np.empty((starts.shape[0], 2), dtype=np.float32)
        seg[:, 0] = starts * self.shift
        seg[:, 1] = ends * self.shift
        if dec[n - 1] != 0:
            end_t = n * self.shift
            if self.flen is not None:
                end_t = end_t + self.flen
            if wav_dur is not None and wav_dur < end_t:
                end_t = wav_dur
            seg[-1, 1] = end_t
        return [(round(s, 3), round(e, 3)) for s, e in seg.tolist()]
