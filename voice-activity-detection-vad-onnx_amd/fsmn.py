"""FSMN-VAD on MI355X: the reference's ORT-session boundary + its sliding-window host loop.

Mirrors FSMN/Inference_FSMN_VAD_ONNX.py (session.run feeds/fetches :170-187, loop :156-234) and the
graph FSMN/Export_FSMN_VAD.py:75-101.  `FsmnSession.run` keeps the named-tensor contract
(`audio`, `cache_0..3`, `one_minus_speech_threshold`, `noise_average_dB` -> `score`, `cache_0..3`,
`noisy_dB`) and is batched over independent streams; `FsmnEngine.detect` runs whole clips on device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import checkpoints as _checkpoints
from . import frontend as _frontend
from . import timestamps as _ts
from . import weights as _weights

SAMPLE_RATE = 16000
OUTPUT_FRAME_LENGTH = 160
PROJ, HIST = 128, 19


class _Meta:
    def __init__(self, name, shape, type_):
        self.name, self.shape, self.type = name, shape, type_


class FsmnEngine:
    def __init__(self, weights=None, input_audio_length=16000, device="cuda:0", speech_2_noise_ratio=1.0):
        torch = _lib.require_gpu()
        self.torch = torch
        self.device = torch.device(device)
        w = _checkpoints.resolve("fsmn", weights)
        w = {k: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in w.items()}
        self.L = int(input_audio_length)
        self.fe = _frontend.Frontend("fsmn", self.L, device=device)
        self.T = self.fe.frames
        dims = _lib.FsmnDims()
        dims.input_affine_dim, dims.linear_dim = w["in1_w"].shape[0], w["in2_w"].shape[0]
        dims.output_affine_dim, dims.output_dim = w["out1_w"].shape[0], w["out2_w"].shape[0]
        dims.frames, dims.speech_2_noise_ratio = self.T, float(speech_2_noise_ratio)
        if w["in1_w"].shape[1] != 400 or w["l0_lin_w"].shape[0] != PROJ or w["l0_fir_w"].shape != (PROJ, 20):
            raise ValueError("FSMN weights: input dim 400, proj 128 and lorder 20 are fixed by the reference cache shape")
        self._dims0 = dims
        hw = _lib.FsmnWeightsHost()
        for k in ("in1_w", "in1_b", "in2_w", "in2_b", "out1_w", "out1_b", "out2_w", "out2_b", "cmvn_means", "cmvn_vars"):
            setattr(hw, k, w[k].ctypes.data)
        for l in range(4):
            hw.lin_w[l], hw.fir_w[l] = w[f"l{l}_lin_w"].ctypes.data, w[f"l{l}_fir_w"].ctypes.data
            hw.aff_w[l], hw.aff_b[l] = w[f"l{l}_aff_w"].ctypes.data, w[f"l{l}_aff_b"].ctypes.data
        Lb = _lib.lib()
        self._hw, self._w = hw, w            # (the host arrays behind hw's pointers stay alive with the engine)

        def build(mode):
            """(dims, device blob) of one arithmetic: the blob carries the weight fragments of that arithmetic only"""
            d = _lib.FsmnDims()
            C.memmove(C.byref(d), C.byref(self._dims0), C.sizeof(d))
            d.arithmetic = _lib.GEMM_MODES[mode]
            n = Lb.vadx_fsmn_packed_floats(C.byref(d))
            if n == 0:
                if mode != "f32":             # dims outside the split tile: float32 MFMAs
                    return build("f32")
                raise ValueError("FSMN dims not supported by the HIP kernel (affine dims <= 144, linear/output <= 256)")
            packed = np.zeros(n, dtype=np.float32)
            _lib.check(Lb.vadx_fsmn_pack_host(C.byref(d), C.byref(hw), packed.ctypes.data))
            return d, torch.from_numpy(packed).to(self.device)

        def flag(d, blob):
            f, a = C.c_uint32(0), C.c_float(0.0)
            with torch.cuda.device(self.device):
                _lib.check(Lb.vadx_fsmn_range_flag(C.byref(d), blob.data_ptr(), 1, C.byref(f), C.byref(a), _lib.stream_ptr()))
            return int(f.value), float(a.value)
        self.blobs = _lib.ArithBlobs(build, flag)
        self.blobs.get()                     # pack now: unsupported dims raise here

    @property
    def dims(self):
        return self.blobs.get()[0]

    @property
    def packed(self):
        return self.blobs.get()[1]

    # ---- front-end + energy for a [B, N] int16 batch cut into W windows at `stride`
    def features(self, audio_i16, windows_per_clip, stride):
        t = self.torch
        a = audio_i16.to(self.device).contiguous()
        B = a.shape[0]
        W = int(windows_per_clip)
        fe = self.fe
        logmel = t.empty((B * W, self.T, 80), dtype=t.float32, device=self.device)
        means = t.empty((B * W,), dtype=t.float32, device=self.device)
        db = t.empty((B * W, self.T), dtype=t.float32, device=self.device)
        L = _lib.lib()
        with t.cuda.device(self.device):
            # window means + frame energies in one pass over the PCM, then the log-mel front-end with those means
            _lib.check(L.vadx_fsmn_window_stats(a.data_ptr(), _lib.row_stride(a), int(stride), B, W, self.L, self.T, means.data_ptr(),
                                                db.data_ptr(), _lib.stream_ptr()))
            _lib.check(L.vadx_frontend_logmel_means(C.byref(fe.cfg), fe.packed.data_ptr(), fe.mel_kb.ctypes.data, a.data_ptr(),
                                                    _lib.row_stride(a), int(stride), B, W, means.data_ptr(), logmel.data_ptr(),
                                                    _lib.stream_ptr()))
        return logmel, db

    def run(self, audio_i16, caches, thr, noise_db, return_psil=False):
        """One boundary call for B streams: audio int16 [B,L]; caches 4 x [B,128,19]; thr, noise_db [B]."""
        t = self.torch
        a = audio_i16.to(self.device)
        B = a.shape[0]
        if a.shape[1] != self.L:
            raise ValueError(f"audio must be [B,{self.L}], got {tuple(a.shape)}")
        logmel, db = self.features(a, 1, self.L)
        cin = [c.to(self.device, t.float32).contiguous() for c in caches]
        for c in cin:
            if tuple(c.shape) != (B, PROJ, HIST):
                raise ValueError(f"cache must be [{B},{PROJ},{HIST}], got {tuple(c.shape)}")
        cout = [t.empty_like(c) for c in cin]
        thr = t.as_tensor(thr, dtype=t.float32, device=self.device).reshape(-1).expand(B).contiguous()
        nz = t.as_tensor(noise_db, dtype=t.float32, device=self.device).reshape(-1).expand(B).contiguous()
        score = t.empty((B, self.T), dtype=t.uint8, device=self.device)
        noisy = t.empty((B,), dtype=t.float32, device=self.device)
        psil = t.empty((B, self.T), dtype=t.float32, device=self.device) if return_psil else None
        pin = (C.c_void_p * 4)(*[c.data_ptr() for c in cin])
        pout = (C.c_void_p * 4)(*[c.data_ptr() for c in cout])

        def launch(mode, dims, packed):
            with t.cuda.device(self.device):
                _lib.check(_lib.lib().vadx_fsmn_run(C.byref(dims), packed.data_ptr(), logmel.data_ptr(), db.data_ptr(),
                                                    C.byref(pin), C.byref(pout), thr.data_ptr(), nz.data_ptr(), B,
                                                    score.data_ptr(), noisy.data_ptr(),
                                                    None if psil is None else psil.data_ptr(), _lib.stream_ptr()))
            return (score, cout, noisy, psil) if return_psil else (score, cout, noisy)
        return self.blobs.guarded(launch)

    # ---- whole clips
    def grid(self, look_backward_s=0.3):
        lb = int(look_backward_s * SAMPLE_RATE // OUTPUT_FRAME_LENGTH)
        stride = self.L - (lb + 1) * OUTPUT_FRAME_LENGTH
        return lb, stride

    def flags(self, padded_i16, windows_per_clip, *, look_backward_s=0.3, speaking_score=0.5, silence_score=0.5,
              snr_threshold=10.0, noise_init_dB=30.0, one_minus_speech_threshold=1.0, return_noise=False):
        """padded_i16 [B, (W-1)*stride + L] int16 on the window grid -> silence flags u8 [B, W*(T-lb)+lb]."""
        t = self.torch
        lb, stride = self.grid(look_backward_s)        # lb may be 0: W*T flags, no tail (Inference_FSMN_VAD_ONNX.py:79-86)
        a = padded_i16.to(self.device).contiguous()
        B, W = a.shape[0], int(windows_per_clip)
        logmel, db = self.features(a, W, stride)
        lp = _lib.FsmnLoopParams()
        lp.look_backward = lb
        lp.one_minus_speech_threshold = float(one_minus_speech_threshold)
        lp.noise_db_init = float(np.float32(noise_init_dB + snr_threshold) * np.float32(0.1))
        lp.snr_threshold = float(snr_threshold * 0.1)
        lp.speaking_score, lp.silence_score = float(speaking_score), float(silence_score)
        nflags = W * (self.T - lb) + lb
        flags = t.empty((B, nflags), dtype=t.uint8, device=self.device)
        cache = t.empty((B, 4, PROJ, HIST), dtype=t.float32, device=self.device)
        trace = t.empty((B, W), dtype=t.float32, device=self.device) if return_noise else None

        def launch(mode, dims, packed):
            with t.cuda.device(self.device):
                _lib.check(_lib.lib().vadx_fsmn_clips(C.byref(dims), packed.data_ptr(), logmel.data_ptr(),
                                                      db.data_ptr(), B, W, C.byref(lp), cache.data_ptr(), flags.data_ptr(),
                                                      None if trace is None else trace.data_ptr(), _lib.stream_ptr()))
            return (flags, trace) if return_noise else flags
        return self.blobs.guarded(launch)

    def flags_from_host(self, host_padded_i16, windows_per_clip, chunk_clips=256, feed=None, **loop_kw):
        """`flags` fed from HOST memory (int16 [B, (W-1)*stride + L], ideally pinned: vadx.feed.pin): chunks of clips are uploaded on
        a copy stream while the previous chunk's launches run (vadx.feed.HostPcmFeed); bit-identical to `flags` of the resident batch."""
        from . import feed as _feed
        f = feed or _feed.HostPcmFeed(self.device, host_padded_i16.shape[1], chunk_clips)
        return _feed.cat_results(f.map([host_padded_i16], lambda a: self.flags(a, windows_per_clip, **loop_kw)))

    def detect(self, clips_i16, pad_noise=None, fusion_threshold=0.3, min_speech_duration=0.2, normalize=True, **loop_kw):
        """Equal-length clips int16 [B,N] (host numpy) -> per clip [(start_s, end_s)], as the reference
        script would print for each.  pad_noise: standard-normal array [B, >=pad] replacing the
        reference's unseeded np.random.normal tail padding (explicit so results are reproducible)."""
        clips = np.asarray(clips_i16)
        B, n = clips.shape
        lb, stride = self.grid(loop_kw.get("look_backward_s", 0.3))
        rows = []
        for b in range(B):
            a = _ts.normalize_to_int16(clips[b].astype(np.float32)) if normalize else clips[b]
            rows.append(pad_to_window_grid(a, self.L, stride, None if pad_noise is None else pad_noise[b]))
        padded = np.stack(rows)
        W = (padded.shape[1] - self.L) // stride + 1
        flags = self.flags(self.torch.from_numpy(padded), W, **loop_kw).cpu().numpy()
        out = []
        for b in range(B):
            ts = _ts.vad_to_timestamps(flags[b].astype(bool), OUTPUT_FRAME_LENGTH / SAMPLE_RATE)
            out.append(_ts.process_timestamps(ts, fusion_threshold, min_speech_duration))
        return out


def pad_to_window_grid(audio_i16, window, stride, noise=None):
    """Tail padding to the sliding-window grid with white noise at the RMS of the tail
    (FSMN/Inference_FSMN_VAD_ONNX.py:88-99).  `noise` = standard-normal samples (seeded by the
    caller); None draws from numpy's global RNG like the reference does."""
    a = np.asarray(audio_i16).reshape(-1)
    n = a.shape[0]
    if n > window:
        num_windows = int(np.ceil((n - window) / stride)) + 1
        pad = (num_windows - 1) * stride + window - n
        if pad == 0:
            return a.copy()
        ref = a[-pad:].astype(np.float32)
    elif n < window:
        pad = window - n
        ref = a.astype(np.float32)
    else:
        return a.copy()
    z = np.random.normal(loc=0.0, scale=1.0, size=pad) if noise is None else np.asarray(noise[:pad], dtype=np.float64)
    fill = (np.sqrt(np.mean(ref * ref)) * z).astype(a.dtype)
    return np.concatenate((a, fill))


class FsmnSession:
    """onnxruntime.InferenceSession look-alike for the FSMN graph (names/dtypes/shapes of
    FSMN/Export_FSMN_VAD.py:115-134); feeds may carry a leading batch of independent streams.

    io_dtype="float16" is the drop-in for the reference's fp16-optimised model (FSMN/Optimize_ONNX.py:48-54 converts with
    keep_io_types=False, so caches, thresholds and noisy_dB cross the boundary as float16 and the driver builds float16
    feeds when `_inputs_meta[1].type` says so, Inference_FSMN_VAD_ONNX.py:40,157-164).  Only the BOUNDARY is half precision
    here: feeds are widened to float32, every kernel computes in float32 (>= the reference's precision), fetches are
    rounded to float16 once.  Against the float32 session fed the same (float16-representable) values the uint8 score
    is identical and caches / noisy_dB differ by that single rounding (<= 2^-11 relative)."""

    def __init__(self, weights=None, input_audio_length=16000, device="cuda:0", io_dtype="float32", speech_2_noise_ratio=1.0):
        if io_dtype not in ("float32", "float16"):
            raise ValueError("io_dtype must be 'float32' or 'float16'")
        self.io_dtype = np.float16 if io_dtype == "float16" else np.float32
        ftype = "tensor(float16)" if io_dtype == "float16" else "tensor(float)"
        self.engine = FsmnEngine(weights, input_audio_length, device, speech_2_noise_ratio)
        L, T = self.engine.L, self.engine.T
        cache = [1, PROJ, HIST, 1]
        self._inputs_meta = [_Meta("audio", [1, 1, L], "tensor(int16)")] + \
            [_Meta(f"cache_{i}", cache, ftype) for i in range(4)] + \
            [_Meta("one_minus_speech_threshold", [1], ftype), _Meta("noise_average_dB", [1], ftype)]
        self._outputs_meta = [_Meta("score", [T], "tensor(uint8)")] + \
            [_Meta(f"cache_{i}", cache, ftype) for i in range(4)] + [_Meta("noisy_dB", [], ftype)]

    def get_inputs(self):
        return list(self._inputs_meta)

    def get_outputs(self):
        return list(self._outputs_meta)

    def get_providers(self):
        return ["VadxMI355XExecutionProvider"]

    def run(self, output_names, feeds):
        t = self.engine.torch
        audio = np.asarray(feeds["audio"])
        if audio.dtype != np.int16:
            raise ValueError("Unexpected input data type. Actual: (%s) , expected: (tensor(int16))" % audio.dtype)
        for name in [f"cache_{i}" for i in range(4)] + ["one_minus_speech_threshold", "noise_average_dB"]:
            got = np.asarray(feeds[name]).dtype
            if got != self.io_dtype:       # onnxruntime refuses a feed of the wrong element type; so does this session
                raise ValueError("Unexpected input data type. Actual: (tensor(%s)) , expected: (tensor(%s))"
                                 % (got, "float16" if self.io_dtype == np.float16 else "float"))
        audio = audio.reshape(-1, audio.shape[-1])
        B = audio.shape[0]
        caches = [t.from_numpy(np.ascontiguousarray(np.asarray(feeds[f"cache_{i}"], dtype=np.float32)).reshape(B, PROJ, HIST))
                  for i in range(4)]
        score, cout, noisy = self.engine.run(t.from_numpy(np.ascontiguousarray(audio)), caches,
                                             np.asarray(feeds["one_minus_speech_threshold"], dtype=np.float32),
                                             np.asarray(feeds["noise_average_dB"], dtype=np.float32))
        io = self.io_dtype
        res = {"score": score.cpu().numpy().reshape(-1) if B == 1 else score.cpu().numpy(),
               "noisy_dB": (noisy.cpu().numpy().reshape(()) if B == 1 else noisy.cpu().numpy()).astype(io)}
        for i in range(4):
            res[f"cache_{i}"] = cout[i].cpu().numpy().reshape(B, PROJ, HIST, 1).astype(io)
        # outputs are name-addressed; the reference script's positional names o0..o5 map in graph order
        order = ["score", "cache_0", "cache_1", "cache_2", "cache_3", "noisy_dB"]
        names = order if output_names is None else output_names
        return [res[n] for n in names]
