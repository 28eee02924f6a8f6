"""VadPostprocessor on device (reference row a16): batched `process` + `decision_to_segment`.

Same constructor arguments and results as the reference classes
(FireRedVAD/Inference_FireRed_ONNX.py:102-304 -- frame shift 0.01 s and +0.025 s frame length at the
end of audio; NVIDIA_.../Inference_NVIDIA_MarbleNet_VAD_ONNX.py:160-353 -- takes frame_shift_s, no
frame-length term).  The decision passes run in libvadx (one clip per thread); only the float32
seconds arithmetic and Python round(,3) of decision_to_segment stay on the host.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class VadPostprocessor:
    def __init__(self, smooth_window_size, prob_threshold, min_speech_frame, max_speech_frame, min_silence_frame,
                 merge_silence_frame, extend_speech_frame, frame_shift_s=0.01, frame_length_s=0.025, device="cuda:0"):
        self.torch = _lib.require_gpu()
        self.device = self.torch.device(device)
        p = _lib.VadPostParams()
        p.smooth_window_size = max(1, int(smooth_window_size))
        p.prob_threshold = float(np.float32(prob_threshold))
        p.min_speech_frame, p.max_speech_frame = int(min_speech_frame), int(max_speech_frame)
        p.min_silence_frame, p.merge_silence_frame = int(min_silence_frame), int(merge_silence_frame)
        p.extend_speech_frame = int(extend_speech_frame)
        self.params = p
        self.frame_shift = np.float32(frame_shift_s)
        self.frame_length = None if frame_length_s is None else np.float32(frame_length_s)

    def process_batch(self, probs, n_frames=None, cap=64):
        """probs f32 [B, S] (device or host), n_frames [B] -> (decisions i8 [B,S], segs i32 [B,cap,2], counts)."""
        t = self.torch
        if not t.is_tensor(probs):
            probs = t.from_numpy(np.ascontiguousarray(probs, dtype=np.float32))
        probs = probs.to(self.device, t.float32).contiguous()
        if probs.dim() == 1:
            probs = probs.unsqueeze(0)
        B, S = probs.shape
        nf = np.full((B,), S, dtype=np.int32) if n_frames is None else np.asarray(n_frames, dtype=np.int32)
        dec = t.zeros((B, max(S, 1)), dtype=t.int8, device=self.device)
        if S == 0:
            return dec[:, :0], t.zeros((B, cap, 2), dtype=t.int32, device=self.device), t.zeros((B,), dtype=t.int32, device=self.device)
        nfd = t.from_numpy(nf).to(self.device)
        L = _lib.lib()
        ws = t.empty(L.vadx_vadpost_workspace_bytes(B, S), dtype=t.uint8, device=self.device)
        while True:
            segs = t.empty((B, cap, 2), dtype=t.int32, device=self.device)
            counts = t.empty((B,), dtype=t.int32, device=self.device)
            with t.cuda.device(self.device):
                _lib.check(L.vadx_vadpost(C.byref(self.params), probs.data_ptr(), S, nfd.data_ptr(), B, dec.data_ptr(),
                                          segs.data_ptr(), counts.data_ptr(), cap, ws.data_ptr(), ws.numel(),
                                          _lib.stream_ptr()))
            worst = int(counts.max().item())
            if worst <= cap:
                return dec, segs, counts
            cap = worst

    def process(self, raw_probs):
        """Single-track reference signature: probabilities -> int8 decision array (host numpy)."""
        p = np.asarray(raw_probs, dtype=np.float32)
        if p.shape[0] == 0:
            return np.empty(0, dtype=np.int8)
        dec, _, _ = self.process_batch(p[None, :])
        return dec[0].cpu().numpy()

    def segments_to_seconds(self, seg_frames, n, wav_dur=None):
        """float32 seconds + round(,3), exactly the tail of decision_to_segment (:169-179)."""
        if len(seg_frames) == 0:
            return []
        seg = np.empty((len(seg_frames), 2), dtype=np.float32)
        fr = np.asarray(seg_frames, dtype=np.float32)
        seg[:, 0] = fr[:, 0] * self.frame_shift
        seg[:, 1] = fr[:, 1] * self.frame_shift
        if int(seg_frames[-1][1]) == n:              # last decision frame is speech
            end_t = n * self.frame_shift
            if self.frame_length is not None:
                end_t = end_t + self.frame_length
            if wav_dur is not None and wav_dur < end_t:
                end_t = wav_dur
            seg[-1, 1] = end_t
        return [(round(s, 3), round(e, 3)) for s, e in seg.tolist()]

    def decision_to_segment(self, decisions, wav_dur=None):
        dec = np.asarray(decisions, dtype=np.int8)
        n = dec.shape[0]
        if n == 0:
            return []
        edges = np.diff(np.concatenate(([0], dec, [0])).astype(np.int8))
        starts, ends = np.flatnonzero(edges == 1), np.flatnonzero(edges == -1)
        return self.segments_to_seconds(list(zip(starts.tolist(), ends.tolist())), n, wav_dur)


class StreamVadPostprocessor:
    """Streaming (frame-by-frame, stateful) decision logic of FireRed Stream-VAD:
    same constructor / `reset` / `process_batch` as FireRedVAD/Export_FireRedVAD.py:1161-1339.
    Host Python like the reference: it is an O(frames) scalar state machine fed by the device probabilities."""

    _SIL, _MAYBE_SP, _SP, _MAYBE_SIL = 0, 1, 2, 3

    def __init__(self, smooth_window_size, speech_threshold, pad_start_frame, min_speech_frame,
                 max_speech_frame, min_silence_frame, frames_per_second=100):
        self.smooth_window_size = max(1, smooth_window_size)
        self.speech_threshold = np.float32(speech_threshold)
        self.pad_start_frame = max(self.smooth_window_size, pad_start_frame)
        self.min_speech_frame, self.max_speech_frame = min_speech_frame, max_speech_frame
        self.min_silence_frame = min_silence_frame
        self.frames_per_second = frames_per_second
        self._window_buf = np.zeros(self.smooth_window_size, dtype=np.float32)
        self.reset()

    def reset(self):
        self._window_buf[:] = 0.0
        self._window_sum = np.float32(0.0)
        self._window_pos = self._window_count = 0
        self.frame_cnt, self.state = 0, self._SIL
        self.speech_cnt = self.silence_cnt = 0
        self.hit_max_speech = False
        self.last_speech_start_frame = self.last_speech_end_frame = -1

    def _smooth(self, p):
        if self.smooth_window_size <= 1:
            return p
        k = self._window_pos
        self._window_sum += p - self._window_buf[k]
        self._window_buf[k] = p
        self._window_pos = (k + 1) % self.smooth_window_size
        if self._window_count < self.smooth_window_size:
            self._window_count += 1
        return self._window_sum / self._window_count

    def _close(self, fc):
        """a segment ends at frame fc: returns its (start, end) and records the end"""
        seg = (self.last_speech_start_frame, fc)
        self.last_speech_start_frame, self.last_speech_end_frame = -1, fc
        return seg

    def process_batch(self, raw_probs):
        probs = raw_probs if isinstance(raw_probs, np.ndarray) else np.asarray(raw_probs, dtype=np.float32)
        if probs.shape[0] == 0:
            return []
        inv_fps = 1.0 / self.frames_per_second
        found = []
        for p in probs:
            self.frame_cnt += 1
            fc = self.frame_cnt
            speech = self._smooth(p) >= self.speech_threshold
            ended = None
            if self.hit_max_speech:                      # a forced split re-opens a segment on the next frame
                self.last_speech_start_frame, self.hit_max_speech = fc, False
            st = self.state
            if st == self._SIL:
                if speech:
                    self.state, self.speech_cnt = self._MAYBE_SP, 1
                else:
                    self.silence_cnt += 1
                    self.speech_cnt = 0
            elif st == self._MAYBE_SP:
                if speech:
                    self.speech_cnt += 1
                    if self.speech_cnt >= self.min_speech_frame:
                        self.state = self._SP
                        self.last_speech_start_frame = max(1, fc - self.speech_cnt + 1 - self.pad_start_frame,
                                                           self.last_speech_end_frame + 1)
                        self.silence_cnt = 0
                else:
                    self.state, self.silence_cnt, self.speech_cnt = self._SIL, 1, 0
            else:
                self.speech_cnt += 1
                if speech:
                    self.state, self.silence_cnt = self._SP, 0
                    if self.speech_cnt >= self.max_speech_frame:
                        self.hit_max_speech, self.speech_cnt = True, 0
                        ended = self._close(fc)
                elif st == self._SP:
                    self.state, self.silence_cnt = self._MAYBE_SIL, 1
                else:
                    self.silence_cnt += 1
                    if self.silence_cnt >= self.min_silence_frame:
                        self.state, self.speech_cnt = self._SIL, 0
                        ended = self._close(fc)
            if ended is not None and ended[0] > 0:
                found.append((max(0, ended[0] - 1) * inv_fps, max(0, ended[1] - 1) * inv_fps))
        if self.last_speech_start_frame > 0:             # unterminated segment at the end of the stream
            found.append((max(0, self.last_speech_start_frame - 1) * inv_fps, (self.frame_cnt - 1) * inv_fps))
        return found


class StreamVadPostprocessorBatch:
    """`StreamVadPostprocessor` for MANY streams at once, on the device: one thread per stream, the state (moving-average ring buffer,
    four-state machine, counters) carried in a device record between `process_batch` calls (include/vadx.h: vadx_stream_vadpost).
    Same constructor arguments as the reference class (FireRedVAD/Export_FireRedVAD.py:1161-1190); `process_batch(track)` takes
    probabilities float32 [streams, frames] (device tensor or array) and returns, per stream, what the reference's process_batch
    returns for that chunk: the (start_s, end_s) of every segment that ended in it, plus the segment still open after its last
    frame (as the reference reports it) -- bit-identical to running the host class per stream."""

    def __init__(self, smooth_window_size, speech_threshold, pad_start_frame, min_speech_frame, max_speech_frame,
                 min_silence_frame, frames_per_second=100, streams=1, device="cuda:0", cap=None):
        import ctypes as C
        self.torch = t = _lib.require_gpu()
        self.device = t.device(device)
        self.streams, self.cap, self.frames_per_second = int(streams), cap, frames_per_second      # cap None: what a chunk can hold at most
        self.prm = _lib.StreamVadPostParams(int(smooth_window_size), float(np.float32(speech_threshold)), int(pad_start_frame),
                                            int(min_speech_frame), int(max_speech_frame), int(min_silence_frame))
        # The device record's ring buffer holds 16 frames: a wider smoothing window runs the host class per stream (same results, the
        # reference's own code path) instead of refusing what the reference accepts.
        self._host = None
        if max(1, int(smooth_window_size)) > 16:
            self._host = [StreamVadPostprocessor(smooth_window_size, speech_threshold, pad_start_frame, min_speech_frame, max_speech_frame,
                                                 min_silence_frame, frames_per_second) for _ in range(self.streams)]
        self._C = C
        self.state = None                  # device record of the streams' state; the host path keeps its state in the host objects
        if self._host is None:
            nbytes = _lib.lib().vadx_stream_vadpost_state_bytes(self.streams)
            self.state = t.zeros(nbytes, dtype=t.uint8, device=self.device)
        self._fresh = True

    def reset(self):
        self._fresh = True
        for h in self._host or ():
            h.reset()

    def process_batch(self, track, flush=True):
        t, C = self.torch, self._C
        if self._host is not None:
            tr = track.detach().to(dtype=t.float32).cpu().numpy() if t.is_tensor(track) else np.asarray(track, dtype=np.float32)
            if tr.ndim != 2 or tr.shape[0] != self.streams:
                raise ValueError(f"track must be [{self.streams}, frames], got {tuple(tr.shape)}")
            out = []
            for s_, h in enumerate(self._host):
                found = h.process_batch(tr[s_])
                # the reference's class always reports the segment still open after the chunk's last frame; flush=False (as the device path's
                # flush flag) leaves it to the chunk that ends it
                if not flush and found and h.last_speech_start_frame > 0:
                    found = found[:-1]
                out.append(found)
            return out
        x = track if t.is_tensor(track) else t.from_numpy(np.ascontiguousarray(track, dtype=np.float32))
        x = x.to(device=self.device, dtype=t.float32)
        if x.dim() != 2 or x.shape[0] != self.streams:
            raise ValueError(f"track must be [{self.streams}, frames], got {tuple(x.shape)}")
        if x.shape[1] == 0:
            return [[] for _ in range(self.streams)]
        if x.stride(1) != 1:
            x = x.contiguous()
        # worst case: max_speech_frame = 1 closes one segment per frame by forced splits; + the open one.  (The kernel advances the device
        # state even when the table overflows, so the table must never be the thing that is too small.)
        cap = max(int(self.cap) if self.cap else 0, int(x.shape[1]) + 1)
        segs = t.empty((self.streams, cap, 2), dtype=t.int32, device=self.device)
        counts = t.empty((self.streams,), dtype=t.int32, device=self.device)
        with t.cuda.device(self.device):
            _lib.check(_lib.lib().vadx_stream_vadpost(C.byref(self.prm), x.data_ptr(), _lib.row_stride(x), int(x.shape[1]), self.streams,
                                                      self.state.data_ptr(), 1 if self._fresh else 0, 1 if flush else 0,
                                                      segs.data_ptr(), counts.data_ptr(), cap, _lib.stream_ptr()))
        self._fresh = False
        cn = counts.cpu().numpy()
        if int(cn.max()) > cap:
            raise _lib.VadxError(f"stream post-processor: {int(cn.max())} segments in one chunk exceed cap={cap}")
        used = int(cn.max())                # the table is sized for the worst case (one segment per frame); only its used prefix travels
        sg = segs[:, :used].cpu().numpy() if used else np.zeros((self.streams, 0, 2), np.int32)
        inv_fps = 1.0 / self.frames_per_second
        return [[(int(a) * inv_fps, int(b) * inv_fps) for a, b in sg[s, :cn[s]].tolist()] for s in range(self.streams)]
