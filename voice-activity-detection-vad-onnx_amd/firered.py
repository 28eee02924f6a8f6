"""FireRedVAD / FireRedAED (non-stream) and FireRed Stream-VAD on MI355X: ORT-session boundary + the
non-overlapping window loop of FireRedVAD/Inference_FireRed_ONNX.py:523-613 (VAD) / :620-742 (AED, odim = 3)
and the cache-carrying chunk loop of :744-822 (Stream-VAD)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import checkpoints as _checkpoints
from . import frontend as _frontend
from . import vadpost as _vadpost
from . import weights as _weights
from .fsmn import _Meta, pad_to_window_grid

SAMPLE_RATE, WINDOW_LENGTH, HOP_LENGTH = 16000, 400, 160
STREAM_CHUNK_SAMPLES = 2560          # 160 ms (Export_FireRedVAD.py:52-53)


def valid_frame_count(num_samples, in_sample_rate=SAMPLE_RATE):
    """snip_edges frame count of a clip of `num_samples` input-rate samples (Inference_FireRed_ONNX.py:84-89)."""
    resampled = int(num_samples * SAMPLE_RATE / in_sample_rate)
    return 0 if resampled < WINDOW_LENGTH else 1 + (resampled - WINDOW_LENGTH) // HOP_LENGTH


class FireRedEngine:
    def __init__(self, weights=None, input_audio_length=16000, device="cuda:0", in_sample_rate=16000):
        """in_sample_rate: the export's IN_SAMPLE_RATE (FireRedVAD/Export_FireRedVAD.py:431-449): windows hold
        `input_audio_length` samples at that rate and the graph resamples them to 16 kHz itself."""
        torch = _lib.require_gpu()
        self.torch = torch
        self.device = torch.device(device)
        w = _checkpoints.resolve("firered", weights)
        c = dict(w["cfg"])
        w = {k: (np.ascontiguousarray(np.asarray(v), dtype=np.float32) if k != "cfg" else v) for k, v in w.items()}
        self.L = int(input_audio_length)
        self.in_sample_rate = int(in_sample_rate)
        self.fe = _frontend.Frontend("firered", self.L, device=device, in_sample_rate=self.in_sample_rate)
        self._fes = {self.L: self.fe}
        self.T = self.fe.frames
        self.odim = c["odim"]
        self.cache_shape = (c["R"], 1, c["P"], (c["N1"] - 1) * c["S1"])
        cfg = _lib.FireRedCfg()
        for k in ("idim", "R", "M", "H", "P", "N1", "S1", "N2", "S2", "odim"):
            setattr(cfg, k, int(c[k]))
        cfg.frames = self.T
        self._cfg0 = cfg
        hw = _lib.FireRedWeightsHost()
        hw.fc1_w, hw.fc1_b, hw.fc2_w, hw.fc2_b = (w[k].ctypes.data for k in ("fc1_w", "fc1_b", "fc2_w", "fc2_b"))
        for r in range(c["R"]):
            hw.fsmn_lb[r] = w[f"fsmn{r}_lb"].ctypes.data
            if c["N2"] > 0:
                hw.fsmn_la[r] = w[f"fsmn{r}_la"].ctypes.data
            if r > 0:
                hw.blk_fc1_w[r], hw.blk_fc1_b[r] = w[f"blk{r}_fc1_w"].ctypes.data, w[f"blk{r}_fc1_b"].ctypes.data
                hw.blk_fc2_w[r] = w[f"blk{r}_fc2_w"].ctypes.data
        for m in range(c["M"]):
            hw.dnn_w[m], hw.dnn_b[m] = w[f"dnn{m}_w"].ctypes.data, w[f"dnn{m}_b"].ctypes.data
        hw.out_w, hw.out_b = w["out_w"].ctypes.data, w["out_b"].ctypes.data
        Lb = _lib.lib()
        self._hw, self._w = hw, w            # (the host arrays behind hw's pointers stay alive with the engine)

        def build(mode):
            """(cfg, device blob) of one arithmetic: the blob carries the weight fragments of that arithmetic only"""
            c = _lib.FireRedCfg.from_buffer_copy(self._cfg0)
            c.arithmetic = _lib.GEMM_MODES[mode]
            n = Lb.vadx_firered_packed_floats(C.byref(c))
            if n == 0:
                if mode != "f32":             # H / P outside the split kernel's shape: float32 MFMAs
                    return build("f32")
                raise ValueError("FireRed config not supported by the HIP kernel (idim 80, H<=256, P<=128, frames<=112, odim<=4)")
            packed = np.zeros(n, dtype=np.float32)
            _lib.check(Lb.vadx_firered_pack_host(C.byref(c), C.byref(hw), packed.ctypes.data))
            return c, torch.from_numpy(packed).to(self.device)

        def flag(c, blob):
            f, a = C.c_uint32(0), C.c_float(0.0)
            with torch.cuda.device(self.device):
                _lib.check(Lb.vadx_firered_range_flag(C.byref(c), blob.data_ptr(), 1, C.byref(f), C.byref(a), _lib.stream_ptr()))
            return int(f.value), float(a.value)
        self.blobs = _lib.ArithBlobs(build, flag)
        self.blobs.get()                     # pack now: an unsupported config raises here

    @property
    def cfg(self):
        return self.blobs.get()[0]

    @property
    def packed(self):
        return self.blobs.get()[1]

    def run(self, audio_i16, windows_per_clip=1):
        """audio int16 [B, W*L] -> probs f32 [B*W, odim, T] (each window stateless, as the reference)."""
        t = self.torch
        if not t.is_tensor(audio_i16):
            audio_i16 = t.from_numpy(np.ascontiguousarray(audio_i16, dtype=np.int16))
        logmel = self.fe.logmel(audio_i16, windows_per_clip, self.L)
        nwin = logmel.shape[0]
        probs = t.empty((nwin, self.odim, self.T), dtype=t.float32, device=self.device)

        def launch(mode, cfg, packed):
            with t.cuda.device(self.device):
                _lib.check(_lib.lib().vadx_firered_run(C.byref(cfg), packed.data_ptr(), logmel.data_ptr(), nwin,
                                                       probs.data_ptr(), _lib.stream_ptr()))
            return probs
        return self.blobs.guarded(launch)

    def run_from_host(self, host_i16, windows_per_clip=1, chunk_clips=256, feed=None):
        """`run` fed from HOST memory (int16 [B, W*L], ideally pinned: vadx.feed.pin), uploads overlapped with compute
        (vadx.feed.HostPcmFeed); bit-identical to `run` of the resident batch."""
        from . import feed as _feed
        f = feed or _feed.HostPcmFeed(self.device, host_i16.shape[1], chunk_clips)
        return _feed.cat_results(f.map([host_i16], lambda a: self.run(a, windows_per_clip)))

    def _frontend_for(self, length):
        fe = self._fes.get(length)
        if fe is None:
            if len(self._fes) >= 8:                      # chunk length + a few tail lengths; keep it bounded
                self._fes.pop(next(k for k in self._fes if k != self.L))
            fe = self._fes[length] = _frontend.Frontend("firered", length, device=self.device, in_sample_rate=self.in_sample_rate)
        return fe

    def new_caches(self, streams=1):
        """zeros [R, streams, P, (N1-1)*S1] (Inference_FireRed_ONNX.py:775)"""
        R, _, P, pad = self.cache_shape
        return self.torch.zeros((R, streams, P, pad), dtype=self.torch.float32, device=self.device)

    def stream_run(self, audio_i16, caches_in):
        """One chunk of `streams` independent streams: audio int16 [streams, n] (400 <= n, any length),
        caches_in f32 [R, streams, P, pad] -> (probs f32 [streams, odim, T], caches_out); T = (n-400)//160+1.
        Look-back only (N2 == 0), as the Stream-VAD checkpoint (Export_FireRedVAD.py:479-612)."""
        t = self.torch
        if not t.is_tensor(audio_i16):
            audio_i16 = t.from_numpy(np.ascontiguousarray(audio_i16, dtype=np.int16))
        audio_i16 = audio_i16.to(self.device)
        S, n = audio_i16.shape
        if n < WINDOW_LENGTH:
            raise ValueError(f"a stream chunk needs at least {WINDOW_LENGTH} samples, got {n}")
        R, _, P, pad = self.cache_shape
        if tuple(caches_in.shape) != (R, S, P, pad) or caches_in.dtype != t.float32:
            raise ValueError(f"caches_in must be float32 {[R, S, P, pad]}, got {list(caches_in.shape)}")
        fe = self._frontend_for(n)
        logmel = fe.logmel(audio_i16, 1, n)
        cfg = _lib.FireRedCfg.from_buffer_copy(self.cfg)
        cfg.frames = fe.frames
        caches_in = caches_in.to(self.device).contiguous()
        caches_out = t.empty_like(caches_in)
        probs = t.empty((S, self.odim, fe.frames), dtype=t.float32, device=self.device)
        with t.cuda.device(self.device):
            _lib.check(_lib.lib().vadx_firered_stream_run(C.byref(cfg), self.packed.data_ptr(), logmel.data_ptr(), S,
                                                          caches_in.data_ptr(), caches_out.data_ptr(), probs.data_ptr(),
                                                          _lib.stream_ptr()))
        return probs, caches_out

    def stream_detect(self, clips_i16, chunk=STREAM_CHUNK_SAMPLES, post=(5, 0.4, 5, 8, 2000, 20), return_probs=False):
        """Equal-length clips int16 [B,N] (host): every clip is one stream, all streams advance one chunk per
        launch (Inference_FireRed_ONNX.py:788-822) -> per clip [(start_s, end_s)]."""
        t = self.torch
        clips = np.asarray(clips_i16)
        B, n = clips.shape
        dev = t.from_numpy(np.ascontiguousarray(clips, dtype=np.int16)).to(self.device)
        caches = self.new_caches(B)
        parts, pos = [], 0
        while pos < n:
            end = min(pos + chunk, n)
            x = dev[:, pos:end]
            if end - pos < WINDOW_LENGTH:
                x = t.nn.functional.pad(x, (0, WINDOW_LENGTH - (end - pos)))
            pr, caches = self.stream_run(x.contiguous(), caches)
            parts.append(pr[:, 0, :])
            pos = end
        nvalid = valid_frame_count(n)
        if not parts or nvalid == 0:
            out, track = [[] for _ in range(B)], np.zeros((B, 0), np.float32)
            return (out, track) if return_probs else out
        track_dev = t.cat(parts, dim=1)[:, :nvalid].contiguous()
        # the decisions of all streams in one launch (vadx_stream_vadpost: one thread per stream), not a host loop over streams
        out = _vadpost.StreamVadPostprocessorBatch(*post, streams=B, device=self.device).process_batch(track_dev)
        track = track_dev.cpu().numpy()
        return (out, track) if return_probs else out

    def detect(self, clips_i16, pad_noise=None, post=(5, 0.4, 20, 2000, 20, 5, 0), return_probs=False):
        """Equal-length clips int16 [B,N] (host) -> per clip [(start_s, end_s)] for output channel 0
        (VAD driver :535-591).  pad_noise: standard-normal [B, >= pad] for the tail padding."""
        clips = np.asarray(clips_i16)
        B, n = clips.shape
        rows = [pad_to_window_grid(clips[b], self.L, self.L, None if pad_noise is None else pad_noise[b]) for b in range(B)]
        padded = np.stack(rows)
        W = padded.shape[1] // self.L
        probs = self.run(padded, W).view(B, W, self.odim, self.T)
        nvalid = valid_frame_count(n, self.in_sample_rate)
        track = probs[:, :, 0, :].reshape(B, W * self.T)[:, :nvalid].contiguous()
        pp = _vadpost.VadPostprocessor(*post, device=self.device)
        if nvalid == 0:
            out = [[] for _ in range(B)]
            return (out, track) if return_probs else out
        dec, segs, counts = pp.process_batch(track)
        segs, counts = segs.cpu().numpy(), counts.cpu().numpy()
        nfr = track.shape[1]     # min(valid frames, W*T): 10 s clips give 980 of the 998 snip-edge frames
        out = [pp.segments_to_seconds(segs[b, :counts[b]].tolist(), nfr, n / self.in_sample_rate) for b in range(B)]
        return (out, track, dec) if return_probs else out


IDX2EVENT = {0: "speech", 1: "singing", 2: "music"}      # Inference_FireRed_ONNX.py:632


def _detect_events(self, clips_i16, pad_noise=None, thresholds=(0.4, 0.5, 0.5), post=(5, 20, 2000, 20, 5, 0),
                   return_probs=False):
    """FireRedAED driver (Inference_FireRed_ONNX.py:620-742): equal-length clips int16 [B,N] through an
    odim == 3 model -> per clip (event2timestamps, event2ratio).  `post` = (smooth, min_event, max_event,
    min_silence, merge_silence, extend); `thresholds` per event in IDX2EVENT order."""
    if self.odim != len(IDX2EVENT):
        raise ValueError(f"the AED driver needs a {len(IDX2EVENT)}-output model, this one has odim = {self.odim}")
    clips = np.asarray(clips_i16)
    B, n = clips.shape
    rows = [pad_to_window_grid(clips[b], self.L, self.L, None if pad_noise is None else pad_noise[b]) for b in range(B)]
    padded = np.stack(rows)
    W = padded.shape[1] // self.L
    probs = self.run(padded, W).view(B, W, self.odim, self.T)
    nvalid = valid_frame_count(n)
    tracks = probs.permute(0, 2, 1, 3).reshape(B, self.odim, W * self.T)[:, :, :nvalid].contiguous()
    nfr = tracks.shape[2]
    out = [({}, {}) for _ in range(B)]
    for idx, event in IDX2EVENT.items():
        thr = thresholds[idx]
        pp = _vadpost.VadPostprocessor(post[0], thr, *post[1:], device=self.device)
        track = tracks[:, idx, :].contiguous()
        if nfr == 0:
            for b in range(B):
                out[b][0][event], out[b][1][event] = [], 0.0
            continue
        _, segs, counts = pp.process_batch(track)
        segs, counts = segs.cpu().numpy(), counts.cpu().numpy()
        ratio = (track >= np.float32(thr)).float().mean(dim=1).cpu().numpy()
        for b in range(B):
            out[b][0][event] = pp.segments_to_seconds(segs[b, :counts[b]].tolist(), nfr, n / SAMPLE_RATE)
            out[b][1][event] = round(float(ratio[b]), 3)
    return (out, tracks) if return_probs else out


FireRedEngine.detect_events = _detect_events


class FireRedSession:
    """onnxruntime.InferenceSession look-alike: {'audio': int16 [1,1,L]} -> [probs f32 [1,odim,T]]
    (FireRedVAD/Export_FireRedVAD.py:785-807); a leading batch of windows is accepted."""

    def __init__(self, weights=None, input_audio_length=16000, device="cuda:0", in_sample_rate=16000):
        self.engine = FireRedEngine(weights, input_audio_length, device, in_sample_rate)
        self._inputs_meta = [_Meta("audio", [1, 1, self.engine.L], "tensor(int16)")]
        self._outputs_meta = [_Meta("probs", [1, self.engine.odim, self.engine.T], "tensor(float)")]

    def get_inputs(self):
        return list(self._inputs_meta)

    def get_outputs(self):
        return list(self._outputs_meta)

    def get_providers(self):
        return ["VadxMI355XExecutionProvider"]

    def run(self, output_names, feeds):
        audio = np.asarray(feeds["audio"])
        if audio.dtype != np.int16:
            raise ValueError("Unexpected input data type. Actual: (%s) , expected: (tensor(int16))" % audio.dtype)
        if audio.shape[-1] != self.engine.L:
            raise ValueError(f"Got invalid dimensions for input: audio, expected last dim {self.engine.L}")
        probs = self.engine.run(audio.reshape(-1, audio.shape[-1])).cpu().numpy()
        return [probs]


class FireRedStreamSession:
    """Stream-VAD session look-alike: {'audio': int16 [1,1,n], 'caches_in': f32 [R,1,P,pad]} ->
    [probs f32 [1,odim,T], caches_out] (FireRedVAD/Export_FireRedVAD.py:848-872; 'audio' axis 2 is dynamic).
    The weights must have no look-ahead filter (cfg N2 == 0)."""

    def __init__(self, weights=None, device="cuda:0"):
        self.engine = FireRedEngine(weights, STREAM_CHUNK_SAMPLES, device)      # None / "" raise in checkpoints.resolve: no silent default
        e = self.engine
        self._inputs_meta = [_Meta("audio", [1, 1, "audio_len"], "tensor(int16)"),
                             _Meta("caches_in", list(e.cache_shape), "tensor(float)")]
        self._outputs_meta = [_Meta("probs", [1, e.odim, "signal_len"], "tensor(float)"),
                              _Meta("caches_out", list(e.cache_shape), "tensor(float)")]

    def get_inputs(self):
        return list(self._inputs_meta)

    def get_outputs(self):
        return list(self._outputs_meta)

    def get_providers(self):
        return ["VadxMI355XExecutionProvider"]

    def run(self, output_names, feeds):
        audio, caches = np.asarray(feeds["audio"]), np.asarray(feeds["caches_in"])
        if audio.dtype != np.int16:
            raise ValueError("Unexpected input data type. Actual: (%s) , expected: (tensor(int16))" % audio.dtype)
        if caches.dtype != np.float32 or caches.shape != self.engine.cache_shape:
            raise ValueError(f"Got invalid dimensions for input: caches_in, expected {list(self.engine.cache_shape)}")
        t = self.engine.torch
        probs, cout = self.engine.stream_run(audio.reshape(1, -1), t.from_numpy(np.ascontiguousarray(caches)))
        outs = {"probs": probs.cpu().numpy(), "caches_out": cout.cpu().numpy()}
        return [outs[k] for k in (output_names or ["probs", "caches_out"])]
