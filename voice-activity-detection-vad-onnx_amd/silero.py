"""Silero-VAD on MI355X: drop-in for the reference's Silero boundary objects.

Mirrors (same names / arguments / error behaviour):
  * `OnnxWrapper`            -- Silero/modeling_modified/utils_vad.py:10-146
  * `load_silero_vad`        -- Silero/modeling_modified/model.py:9-41
  * `get_speech_timestamps`  -- Silero/modeling_modified/utils_vad.py:248-491
plus the batched entry points the reference does not have (`SileroEngine.clips`,
`get_speech_timestamps_batch`).  All arithmetic runs in libvadx.so (HIP, gfx950); this file is
plumbing: tensors in, C-ABI call, tensors out.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings

import numpy as np

from . import _lib
from . import weights as _weights

CONTEXT_SIZE = 64
NUM_SAMPLES = 512
HIDDEN = 128


def _as_weight_dict(path_or_weights):
    """dict | silero_vad.onnx | .npz with the keys of weights.silero_synthetic() | the explicit opt-in "synthetic:<seed>".
    None / "" raises: the reference's default (the silero_vad package's bundled model, model.py:9-41) does not exist here
    and random weights must never be a silent default (checkpoints.resolve)."""
    from . import checkpoints
    w = checkpoints.resolve("silero", path_or_weights)
    w = {k: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in w.items()}
    _weights.silero_check(w)
    return w


WORKSPACE_CAP_BYTES = 8 << 30     # clips(): above this the gate-preactivation workspace is reused span by span


ENCODER_MODES = {"f32": _lib.ARITH["f32"], "split": _lib.ARITH["split"], "h2": _lib.ARITH["h2"]}      # name -> VADX_ARITH_*
_default_mode = [None]


def encoder_mode(mode=None):
    """The kernel set engines use unless told otherwise per engine (`SileroEngine.arithmetic`): "f32" = exact-f32 MFMAs, "split" =
    bf16 x 3 exact-split products, "h2" = fp16 x 2 split products (float32-class accuracy at 3/16 of the matrix time; activations
    outside the fp16 range are detected and the batch recomputed on "split").  A Python-side default only -- the C ABI takes the
    arithmetic with every call (include/vadx.h: vadx_silero_cfg).  Returns the mode that was active; `None` only queries.  The initial
    value comes from VADX_SILERO_ENCODER, read once here."""
    if _default_mode[0] is None:
        e = os.environ.get("VADX_SILERO_ENCODER", "").strip().lower()
        _default_mode[0] = {"0": "f32", "1": "split", "2": "h2", "bf16x3": "split", "f16x2": "h2"}.get(e, e) if e else "h2"
        if _default_mode[0] not in ENCODER_MODES:
            raise ValueError(f"VADX_SILERO_ENCODER must be one of {sorted(ENCODER_MODES)}, got {e!r}")
    prev = _default_mode[0]
    if mode is not None:
        if mode not in ENCODER_MODES:
            raise ValueError(f"encoder mode must be one of {sorted(ENCODER_MODES)}, got {mode!r}")
        _default_mode[0] = mode
    return prev


class SileroEngine:
    """Device-resident packed weights + workspace; thin wrappers over the C ABI."""

    def __init__(self, path_or_weights=None, device="cuda:0"):
        torch = _lib.require_gpu()
        self.torch = torch
        self.device = torch.device(device)
        L = _lib.lib()
        w = _as_weight_dict(path_or_weights)
        hw = _lib.SileroWeightsHost()
        keep = []

        def ptr(a):
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)

        hw.stft_basis = ptr(w["stft_basis"])
        for i in range(4):
            hw.enc_w[i] = ptr(w[f"enc{i}_w"]).value
            hw.enc_b[i] = ptr(w[f"enc{i}_b"]).value
        hw.lstm_w_ih, hw.lstm_w_hh = ptr(w["lstm_w_ih"]), ptr(w["lstm_w_hh"])
        hw.lstm_b_ih, hw.lstm_b_hh = ptr(w["lstm_b_ih"]), ptr(w["lstm_b_hh"])
        hw.dec_w, hw.dec_b = ptr(w["dec_w"]), ptr(w["dec_b"])
        packed = np.zeros(L.vadx_silero_packed_floats(), dtype=np.float32)
        _lib.check(L.vadx_silero_pack_host(C.byref(hw), packed.ctypes.data_as(C.c_void_p)))
        self.packed = torch.from_numpy(packed).to(self.device)
        self._ws = None
        self.arithmetic = None            # None = the module default (encoder_mode()); or "f32" | "split" | "h2" for this engine
        self.h2_ok = bool(packed[L.vadx_silero_packed_floats() - 4] != 0.0)      # the blob's fp16 x 2 section is usable (pack-time check)
        self.range_fallbacks = 0          # batches recomputed on bf16 x 3 because an activation left the fp16 range

    # -- arithmetic selection + the fp16 x 2 range protocol
    def mode(self):
        m = self.arithmetic or encoder_mode()
        return "split" if (m == "h2" and not self.h2_ok) else m

    def cfg(self, mode=None):
        """ctypes pointer to a vadx_silero_cfg for `mode` (default: this engine's current mode)"""
        c = _lib.SileroCfg()
        c.arithmetic = ENCODER_MODES[mode or self.mode()]
        return C.byref(c)

    def range_flag(self, reset=True):
        """(flag, largest |activation|) of the fp16 x 2 kernels since the last reset; synchronises the stream (vadx_silero_range_flag)"""
        flag, amax = C.c_uint32(0), C.c_float(0.0)
        with self.torch.cuda.device(self.device):
            _lib.check(_lib.lib().vadx_silero_range_flag(self.packed.data_ptr(), 1 if reset else 0, C.byref(flag), C.byref(amax),
                                                         _lib.stream_ptr()))
        return int(flag.value), float(amax.value)

    def _guarded(self, run):
        """run(mode) -> result.  On "h2" the result stands only if no activation left the fp16 range; otherwise the batch is recomputed on
        "split" (bf16 terms have float32's range).  One 8-byte read-back + stream synchronisation per guarded call."""
        m = self.mode()
        out = run(m)
        if m == "h2":
            flag, amax = self.range_flag()
            if flag:
                self.range_fallbacks += 1
                out = run("split")
        return out

    def _check_workspace_cap(self, batch, steps, who):
        """The whole-batch gx workspace is 32 KB per 16-clip group and window; `clips` falls back to spans above WORKSPACE_CAP_BYTES,
        the int16 / host-feed entry points have no spanned variant and say so instead of running the allocator out of memory."""
        need = _lib.lib().vadx_silero_workspace_bytes(int(batch), int(steps))
        if need > WORKSPACE_CAP_BYTES:
            raise ValueError(f"{who}: a batch of {batch} clips x {steps} windows needs a {need / 2**30:.1f} GiB workspace "
                             f"(cap {WORKSPACE_CAP_BYTES / 2**30:.0f} GiB): split the batch into groups of at most "
                             f"{max(16, WORKSPACE_CAP_BYTES // _lib.lib().vadx_silero_workspace_bytes(16, int(steps)) * 16)} clips, "
                             "or use clips() (float32, spanned automatically)")

    # -- scratch (gx tiles) grows on demand and is reused
    def _workspace(self, batch, steps):
        need = _lib.lib().vadx_silero_workspace_bytes(int(batch), int(steps))
        if self._ws is None or self._ws.numel() < need:
            self._ws = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
        return self._ws

    def _dev_f32(self, x):
        t = self.torch
        if not t.is_tensor(x):
            x = t.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return x.to(device=self.device, dtype=t.float32).contiguous()

    def step(self, x, state):
        """session.run equivalent: x [B,576], state [2,B,128] -> (out [B,1], stateN [2,B,128]) on device."""
        t = self.torch
        x = self._dev_f32(x)
        state = self._dev_f32(state)
        if x.dim() != 2 or x.shape[1] != CONTEXT_SIZE + NUM_SAMPLES:
            raise ValueError(f"input must be [B,{CONTEXT_SIZE + NUM_SAMPLES}], got {tuple(x.shape)}")
        B = x.shape[0]
        if tuple(state.shape) != (2, B, HIDDEN):
            raise ValueError(f"state must be [2,{B},{HIDDEN}], got {tuple(state.shape)}")
        out = t.empty((B, 1), dtype=t.float32, device=self.device)
        state_n = t.empty_like(state)
        ws = self._workspace(B, 1)

        def run(mode):
            with t.cuda.device(self.device):
                _lib.check(_lib.lib().vadx_silero_step(self.packed.data_ptr(), x.data_ptr(), state.data_ptr(), 16000, B,
                                                       out.data_ptr(), state_n.data_ptr(), ws.data_ptr(), ws.numel(),
                                                       _lib.stream_ptr(), self.cfg(mode)))
            return out, state_n
        return self._guarded(run)

    def clips(self, audio, n_samples=None, return_state=False):
        """audio f32 [B,N] (+-1 scale) -> probs [B, ceil(n/512)] on device (zero state/context at t=0)."""
        t = self.torch
        audio = self._dev_f32(audio)
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        if audio.dim() != 2:
            raise ValueError(f"Too many dimensions for input audio {audio.dim()}")
        B, N = audio.shape
        n = int(N if n_samples is None else n_samples)
        if n <= 0 or n > N:
            raise ValueError(f"n_samples={n} outside (0,{N}]")
        steps = (n + NUM_SAMPLES - 1) // NUM_SAMPLES
        probs = t.empty((B, steps), dtype=t.float32, device=self.device)
        state_n = t.empty((2, B, HIDDEN), dtype=t.float32, device=self.device) if return_state else None
        if _lib.lib().vadx_silero_workspace_bytes(B, steps) > WORKSPACE_CAP_BYTES:
            state_n = self.clips_spanned(audio, n, probs, state_n)
            return (probs, state_n) if return_state else probs
        ws = self._workspace(B, steps)

        def run(mode):
            with t.cuda.device(self.device):
                _lib.check(_lib.lib().vadx_silero_clips(self.packed.data_ptr(), audio.data_ptr(), B, n, _lib.row_stride(audio),
                                                        probs.data_ptr(), None if state_n is None else state_n.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), _lib.stream_ptr(), self.cfg(mode)))
            return (probs, state_n) if return_state else probs
        return self._guarded(run)

    def clips_spanned(self, audio, n, probs, state=None, span=None):
        """`clips` for recordings whose whole-clip workspace (32 KB per 16-clip group and window) would not fit: the
        encoder and the recurrent kernel run span by span over windows [first, first + span) on the caller's stream,
        reusing one span-sized workspace, the LSTM state carried in `state` [2,B,128].  Same kernels and arithmetic
        order as `clips`: identical results.  audio f32 [B,N] on the device; probs [B,steps] is filled."""
        t = self.torch
        L = _lib.lib()
        B = int(audio.shape[0])
        steps = (int(n) + NUM_SAMPLES - 1) // NUM_SAMPLES
        tile_bytes = L.vadx_silero_workspace_bytes(B, 1)
        if span is None:
            span = max(1, min(steps, WORKSPACE_CAP_BYTES // tile_bytes))
        if state is None:
            state = t.empty((2, B, HIDDEN), dtype=t.float32, device=self.device)
        ws = self._workspace(B, min(span, steps))

        def run(mode):
            with t.cuda.device(self.device):
                st = _lib.stream_ptr()
                for first in range(0, steps, span):
                    ns = min(span, steps - first)
                    _lib.check(L.vadx_silero_encode_span(self.packed.data_ptr(), audio.data_ptr(), B, int(n), _lib.row_stride(audio),
                                                         first, ns, ws.data_ptr(), ns * tile_bytes, st, self.cfg(mode)))
                    _lib.check(L.vadx_silero_recur_span(self.packed.data_ptr(), ws.data_ptr(), ns * tile_bytes, B, ns,
                                                        None if first == 0 else state.data_ptr(), probs.data_ptr() + 4 * first,
                                                        steps, state.data_ptr(), st, self.cfg(mode)))
            return state
        return self._guarded(run)

    def encode(self, audio, n_samples=None, mode=None):
        """First half of `clips` as its own launch (fills the workspace); returns (batch, steps).  (The separate halves do not run the
        fp16 x 2 range protocol themselves: pair them through `clips*`, or call `range_flag()` before trusting an "h2" result.)"""
        t = self.torch
        B, N = audio.shape
        n = int(N if n_samples is None else n_samples)
        steps = (n + NUM_SAMPLES - 1) // NUM_SAMPLES
        ws = self._workspace(B, steps)
        with t.cuda.device(self.device):
            _lib.check(_lib.lib().vadx_silero_encode(self.packed.data_ptr(), audio.data_ptr(), B, n, _lib.row_stride(audio),
                                                     ws.data_ptr(), ws.numel(), _lib.stream_ptr(), self.cfg(mode)))
        return B, steps

    PCM16_SCALE = 0.000030517578       # Silero/Inference_Silero_VAD_ONNX.py:83: float32 = int16 * this

    def encode_pcm16(self, pcm, n_samples=None, scale=PCM16_SCALE, mode=None):
        """`encode` fed the int16 samples themselves (device tensor [B,N]): the kernel applies the reference's
        int16 -> float32 scaling while staging, bit-identical to encoding `pcm.float() * float32(scale)`."""
        t = self.torch
        if pcm.dtype != t.int16 or pcm.dim() != 2 or not pcm.is_cuda:
            raise ValueError("encode_pcm16 expects a device int16 tensor [B,N]")
        B, N = pcm.shape
        n = int(N if n_samples is None else n_samples)
        steps = (n + NUM_SAMPLES - 1) // NUM_SAMPLES
        self._check_workspace_cap(B, steps, "encode_pcm16")
        ws = self._workspace(B, steps)
        with t.cuda.device(self.device):
            _lib.check(_lib.lib().vadx_silero_encode_pcm16(self.packed.data_ptr(), pcm.data_ptr(), float(scale), B, n,
                                                           _lib.row_stride(pcm), ws.data_ptr(), ws.numel(), _lib.stream_ptr(),
                                                           self.cfg(mode)))
        return B, steps

    def clips_pcm16(self, pcm, n_samples=None, scale=PCM16_SCALE):
        """int16 PCM [B,N] on the device -> probs [B, ceil(n/512)] (encode_pcm16 + recur)."""
        def run(mode):
            B, steps = self.encode_pcm16(pcm, n_samples, scale, mode=mode)
            return self.recur(B, steps, self.torch.empty((B, steps), dtype=self.torch.float32, device=self.device), mode=mode)
        return self._guarded(run)

    def host_feed(self, batch, n_samples, chunk_clips=512):
        """A reusable upload pipeline for `batch` clips of `n_samples` int16 samples held in HOST memory (the host link,
        not the kernels, bounds an 8-GPU node) -- see HostFeed."""
        return HostFeed(self, batch, n_samples, chunk_clips)

    def clips_from_host(self, host_pcm, chunk_clips=512, feed=None):
        """int16 PCM [B,N] in host memory (numpy or a CPU tensor; pinned memory makes the copies asynchronous) -> probs [B, T]
        on the device, the upload double-buffered against the encoder.  Same scores as `clips_pcm16` of the uploaded batch,
        bit for bit.  Pass a `host_feed(...)` object to reuse its device buffers across calls."""
        t = self.torch
        host = host_pcm if t.is_tensor(host_pcm) else t.from_numpy(np.ascontiguousarray(host_pcm, dtype=np.int16))
        if host.dtype != t.int16 or host.dim() != 2 or host.is_cuda:
            raise ValueError("clips_from_host expects host int16 PCM [B,N]")
        B, N = host.shape
        if feed is None:
            feed = HostFeed(self, B, N, chunk_clips)
        elif (feed.B, feed.N) != (B, N) or feed.eng is not self:
            raise ValueError("clips_from_host: the feed was built for another engine or batch shape")
        def run(mode):
            feed.encode(host, mode=mode)
            return self.recur(B, feed.T, t.empty((B, feed.T), dtype=t.float32, device=self.device), mode=mode)
        return self._guarded(run)

    def recur(self, batch, steps, probs, mode=None):
        """Second half of `clips`: workspace -> probs [B,steps] (zero initial state)."""
        t = self.torch
        ws = self._workspace(batch, steps)
        with t.cuda.device(self.device):
            _lib.check(_lib.lib().vadx_silero_recur(self.packed.data_ptr(), ws.data_ptr(), ws.numel(), batch, steps,
                                                    None, probs.data_ptr(), None, _lib.stream_ptr(), self.cfg(mode)))
        return probs

    def segments(self, probs, n_samples, cap=64, **kw):
        """Device segmenter: probs [B,T] -> (int64 [B,cap,2] sample indices, int32 [B] counts)."""
        t = self.torch
        probs = self._dev_f32(probs)
        B, T = probs.shape
        if t.is_tensor(n_samples):             # clip lengths may already live on the device
            lens = n_samples.to(device=self.device, dtype=t.int64).reshape(-1).expand(B).contiguous()
        else:
            lens = t.as_tensor(np.broadcast_to(np.asarray(n_samples, dtype=np.int64), (B,)).copy(), device=self.device)
        prm = seg_params(**kw)
        while True:
            segs = t.empty((B, cap, 2), dtype=t.int64, device=self.device)
            counts = t.empty((B,), dtype=t.int32, device=self.device)
            with t.cuda.device(self.device):
                _lib.check(_lib.lib().vadx_silero_segments(probs.data_ptr(), B, T, lens.data_ptr(), C.byref(prm),
                                                           segs.data_ptr(), counts.data_ptr(), cap, _lib.stream_ptr()))
            worst = int(counts.max().item())
            if worst <= cap:
                return segs, counts
            cap = worst          # rare: a clip produced more segments than the table holds -> rerun


class HostFeed:
    """Pinned-host -> device feed of the batched encoder.  Chunks of `chunk_clips` clips cross PCIe on a copy
    stream into one of two device int16 buffers while the encoder of the previous chunk runs on the caller's stream
    (`vadx_silero_encode_pcm16_part` fills the chunk's slice of the ONE batch workspace); the recurrent kernel then runs once
    over the whole batch.  int16 is what a host has (wav files): the encoder applies the reference's x 0.000030517578 itself,
    bit-identically (Silero/Inference_Silero_VAD_ONNX.py:83).  Measured on one MI355X, 4096 x 10 s: 23.7 ms per batch against
    23.0 ms for the upload alone (57 GB/s of PCIe Gen5 x16) -- the compute is hidden completely."""

    def __init__(self, engine, batch, n_samples, chunk_clips=512):
        t = engine.torch
        self.eng, self.B, self.N = engine, int(batch), int(n_samples)
        self.T = (self.N + NUM_SAMPLES - 1) // NUM_SAMPLES
        engine._check_workspace_cap(self.B, self.T, "HostFeed")
        self.chunk = max(16, (int(chunk_clips) + 15) // 16 * 16)       # slices of the workspace start on a 16-clip group boundary
        self.buf = [t.empty((self.chunk, self.N), dtype=t.int16, device=engine.device) for _ in range(2)]
        self.copy_stream = t.cuda.Stream(device=engine.device)
        self.ready = [t.cuda.Event() for _ in range(2)]
        self.free = [t.cuda.Event() for _ in range(2)]
        self.in_use = [False, False]

    def encode(self, host_pcm, mode=None):
        """Upload + encode every chunk of host_pcm [B,N] (int16 CPU tensor); afterwards the engine's workspace holds the batch
        (call `engine.recur(B, T, probs)` next, on the same stream)."""
        eng, t = self.eng, self.eng.torch
        if tuple(host_pcm.shape) != (self.B, self.N) or host_pcm.dtype != t.int16 or host_pcm.is_cuda:
            raise ValueError(f"HostFeed.encode expects host int16 PCM [{self.B},{self.N}]")
        ws = eng._workspace(self.B, self.T)
        L = _lib.lib()
        with t.cuda.device(eng.device):
            comp = t.cuda.current_stream()
            st = _lib.stream_ptr()
            for k2, b0 in enumerate(range(0, self.B, self.chunk)):
                nb = min(self.chunk, self.B - b0)
                k = k2 & 1
                with t.cuda.stream(self.copy_stream):
                    if self.in_use[k]:                  # the encoder launch that last read this buffer (also across calls)
                        self.copy_stream.wait_event(self.free[k])
                    self.buf[k][:nb].copy_(host_pcm[b0:b0 + nb], non_blocking=True)
                    self.ready[k].record(self.copy_stream)
                comp.wait_event(self.ready[k])
                _lib.check(L.vadx_silero_encode_pcm16_part(eng.packed.data_ptr(), self.buf[k].data_ptr(), eng.PCM16_SCALE, nb, self.N,
                                                           self.N, b0, self.B, ws.data_ptr(), ws.numel(), st, eng.cfg(mode)))
                self.free[k].record(comp)
                self.in_use[k] = True
        return self.B, self.T


def seg_params(threshold=0.5, sampling_rate=16000, min_speech_duration_ms=250,
               max_speech_duration_s=float("inf"), min_silence_duration_ms=100, speech_pad_ms=30,
               neg_threshold=None, min_silence_at_max_speech=98, use_max_poss_sil_at_max_speech=True, **_ignored):
    p = _lib.SileroSegParams()
    p.threshold = float(threshold)
    p.neg_threshold = -1.0 if neg_threshold is None else float(neg_threshold)
    p.sampling_rate = int(sampling_rate)
    p.min_speech_duration_ms = float(min_speech_duration_ms)
    p.max_speech_duration_s = float(max_speech_duration_s)
    p.min_silence_duration_ms = float(min_silence_duration_ms)
    p.speech_pad_ms = float(speech_pad_ms)
    p.min_silence_at_max_speech = float(min_silence_at_max_speech)
    p.use_max_poss_sil_at_max_speech = 1 if use_max_poss_sil_at_max_speech else 0
    return p


class SileroSession:
    """What the reference's OnnxWrapper holds as `self.session`: run(None, {'input' [B, ctx + n], 'state' [2,B,128], 'sr' int64})
    -> [out [B,1], stateN [2,B,128]] (utils_vad.py:116-119).  The 16 kHz sub-graph (576-sample input) runs on the HIP engine;
    the upstream file's 8 kHz sub-graph (288-sample input, 128-point STFT, 65-channel first conv) is a different network
    whose weights only exist inside the un-vendored silero_vad.onnx -- it is not built, and asking for it fails loudly."""

    def __init__(self, engine):
        self.engine = engine

    def run(self, output_names, feeds):
        sr = int(np.asarray(feeds["sr"]))
        if sr != 16000:
            raise ValueError(f"sr={sr}: only the 16 kHz sub-graph of the Silero network is built on the HIP path "
                             "(the wrapper and the segmenter handle 8000 Hz; the 8 kHz network itself is not implemented)")
        out, state = self.engine.step(feeds["input"], feeds["state"])
        return [out, state]


class OnnxWrapper:
    """Drop-in for the reference wrapper (utils_vad.py:10-146): validation, state / context carry, reset rules."""

    def __init__(self, path=None, force_onnx_cpu=True, device="cuda:0"):
        self.engine = path if isinstance(path, SileroEngine) else SileroEngine(path, device)
        self.torch = self.engine.torch
        self.session = SileroSession(self.engine)
        self.reset_states()
        if isinstance(path, str) and "16k" in path:
            warnings.warn("This model support only 16000 sampling rate!")
            self.sample_rates = [16000]
        else:
            self.sample_rates = [8000, 16000]        # utils_vad.py:63-67; an 8000 Hz call reaches SileroSession.run, which refuses it

    def _validate_input(self, x, sr: int):
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
        t = self.torch
        self._state = t.zeros((2, batch_size, HIDDEN), dtype=t.float32, device=self.engine.device)
        self._context = t.zeros(0)
        self._last_sr = 0
        self._last_batch_size = 0

    def __call__(self, x, sr: int):
        t = self.torch
        if not t.is_tensor(x):
            x = t.as_tensor(np.asarray(x, dtype=np.float32))
        x, sr = self._validate_input(x, sr)
        num_samples = NUM_SAMPLES if sr == 16000 else 256
        if x.shape[-1] != num_samples:
            raise ValueError(f"Provided number of samples is {x.shape[-1]} "
                             "(Supported values: 256 for 8000 sample rate, 512 for 16000)")
        batch_size = x.shape[0]
        context_size = CONTEXT_SIZE if sr == 16000 else 32
        if not self._last_batch_size:
            self.reset_states(batch_size)
        if self._last_sr and self._last_sr != sr:
            self.reset_states(batch_size)
        if self._last_batch_size and self._last_batch_size != batch_size:
            self.reset_states(batch_size)
        x = x.to(device=self.engine.device, dtype=t.float32)
        if not len(self._context):
            self._context = t.zeros(batch_size, context_size, device=self.engine.device)
        x = t.cat([self._context, x], dim=1)
        out, state = self.session.run(None, {"input": x, "state": self._state, "sr": np.array(sr, dtype="int64")})
        self._state = state if t.is_tensor(state) else t.as_tensor(np.asarray(state), device=self.engine.device)
        self._context = x[..., -context_size:]
        self._last_sr = sr
        self._last_batch_size = batch_size
        return out.cpu() if t.is_tensor(out) else t.as_tensor(np.asarray(out))

    def audio_forward(self, x, sr: int):
        """Whole clips in ONE device call at 16 kHz (the reference loops window by window, utils_vad.py:130-146; at 8000 Hz
        this does too, through `__call__`, so a substituted session sees exactly the reference's feeds)."""
        t = self.torch
        if not t.is_tensor(x):
            x = t.as_tensor(np.asarray(x, dtype=np.float32))
        x, sr = self._validate_input(x, sr)
        self.reset_states()
        if sr != 16000:
            num_samples = 256
            if x.shape[1] % num_samples:
                x = t.nn.functional.pad(x, (0, num_samples - (x.shape[1] % num_samples)), "constant", value=0.0)
            return t.cat([self(x[:, i:i + num_samples], sr) for i in range(0, x.shape[1], num_samples)], dim=1).cpu()
        probs, state = self.engine.clips(x, return_state=True)
        # leave the wrapper exactly where the reference's loop would: last state + last 64 samples
        pad = (-x.shape[1]) % NUM_SAMPLES
        xp = t.nn.functional.pad(x.to(self.engine.device, t.float32), (0, pad))
        self._state, self._context = state, xp[:, -CONTEXT_SIZE:]
        self._last_sr, self._last_batch_size = sr, x.shape[0]
        return probs.cpu()


def segments_from_probs(engine, probs, lengths, **kw):
    """The segmenter half of get_speech_timestamps on given window probabilities (device state machine, utils_vad.py:374-482):
    probs [B, T] (T windows of 512 samples at 16 kHz / 256 at 8 kHz), lengths [B] in samples -> list of per-clip lists of
    {'start','end'}.  kw as get_speech_timestamps (threshold, sampling_rate, min_speech_duration_ms, ..., return_seconds)."""
    engine = engine.engine if isinstance(engine, OnnxWrapper) else engine
    return_seconds = kw.pop("return_seconds", False)
    time_resolution = kw.pop("time_resolution", 1)
    sr = kw.get("sampling_rate", 16000)
    if sr not in (8000, 16000):
        raise ValueError("Currently silero VAD models support 8000 and 16000 (or multiply of 16000) sample rates")
    lens = np.asarray(lengths, dtype=np.int64).reshape(-1)
    segs, counts = engine.segments(probs, lens, **kw)
    return _finish(segs, counts, lens, sr, return_seconds, time_resolution, 1)


def load_silero_vad(onnx=True, opset_version=16, use_cpu=True, path="", device="cuda:0"):
    """Constructor of the boundary object (reference signature; `use_cpu` is accepted and ignored: this build runs on the
    MI355X only).  `path` = the silero_vad.onnx the reference would hand to onnxruntime (its initialisers are read by
    vadx.onnx_reader), a .npz of arrays, or "synthetic:<seed>"; the reference's `path=""` default (the pip package's bundled
    file) cannot be honoured here and raises."""
    if onnx and opset_version not in (15, 16):
        raise Exception("Available ONNX opset_version: [15, 16]")
    return OnnxWrapper(path or None, force_onnx_cpu=use_cpu, device=device)


def _finish(segs, counts, lengths, sampling_rate, return_seconds, time_resolution, step):
    out = []
    segs = segs.cpu().numpy()
    counts = counts.cpu().numpy()
    for b in range(segs.shape[0]):
        row = []
        for s, e in segs[b, :counts[b]].tolist():
            if return_seconds:
                dur = int(lengths[b]) / sampling_rate
                row.append({"start": max(round(s / sampling_rate, time_resolution), 0),
                            "end": min(round(e / sampling_rate, time_resolution), dur)})
            elif step > 1:
                row.append({"start": s * step, "end": e * step})
            else:
                row.append({"start": s, "end": e})
        out.append(row)
    return out


def get_speech_timestamps_batch(audio, model, lengths=None, threshold: float = 0.5, sampling_rate: int = 16000,
                                min_speech_duration_ms: int = 250, max_speech_duration_s: float = float("inf"),
                                min_silence_duration_ms: int = 100, speech_pad_ms: int = 30,
                                return_seconds: bool = False, time_resolution: int = 1,
                                neg_threshold: float = None, min_silence_at_max_speech: int = 98,
                                use_max_poss_sil_at_max_speech: bool = True, return_probs: bool = False):
    """Batched get_speech_timestamps: audio f32 [B,N] (equal length, or `lengths` per clip with
    zero padding beyond) -> list (per clip) of lists of {'start','end'} dicts."""
    engine = model.engine if isinstance(model, OnnxWrapper) else model
    t = engine.torch
    audio = engine._dev_f32(audio)
    if audio.dim() == 1:
        audio = audio.unsqueeze(0)
    step = 1
    if sampling_rate > 16000 and sampling_rate % 16000 == 0:
        step = sampling_rate // 16000
        sampling_rate = 16000
        audio = audio[:, ::step].contiguous()
        warnings.warn("Sampling rate is a multiply of 16000, casting to 16000 manually!")
    if sampling_rate not in (8000, 16000):
        raise ValueError("Currently silero VAD models support 8000 and 16000 (or multiply of 16000) sample rates")
    if sampling_rate == 8000:
        raise ValueError("sampling_rate=8000: only the 16 kHz sub-graph of the Silero network is built on the HIP path; "
                         "window probabilities from elsewhere can be segmented with segments_from_probs(..., sampling_rate=8000)")
    B, N = audio.shape
    if N == 0:           # empty audio: the reference's chunk loop never runs and it returns [] (utils_vad.py:330-344)
        res = [[] for _ in range(B)]
        return (res, t.empty((B, 0), dtype=t.float32, device=engine.device)) if return_probs else res
    if lengths is None:
        lens = np.full((B,), N, dtype=np.int64)
    else:
        lens = np.asarray(lengths, dtype=np.int64)
        if step > 1:
            lens = (lens + step - 1) // step
        if lens.shape != (B,) or lens.min() <= 0 or lens.max() > N:
            raise ValueError("lengths must be B positive values <= audio.shape[1]")
        if lens.min() != N:      # zero everything past each clip's end (the reference zero-pads the last window)
            mask = t.arange(N, device=engine.device).unsqueeze(0) < t.as_tensor(lens, device=engine.device).unsqueeze(1)
            audio = audio * mask
    probs = engine.clips(audio, n_samples=int(lens.max()))
    segs, counts = engine.segments(probs, lens, threshold=threshold, sampling_rate=sampling_rate,
                                   min_speech_duration_ms=min_speech_duration_ms,
                                   max_speech_duration_s=max_speech_duration_s,
                                   min_silence_duration_ms=min_silence_duration_ms, speech_pad_ms=speech_pad_ms,
                                   neg_threshold=neg_threshold, min_silence_at_max_speech=min_silence_at_max_speech,
                                   use_max_poss_sil_at_max_speech=use_max_poss_sil_at_max_speech)
    res = _finish(segs, counts, lens, sampling_rate, return_seconds, time_resolution, step)
    return (res, probs) if return_probs else res


def get_speech_timestamps(audio, model, threshold: float = 0.5, sampling_rate: int = 16000,
                          min_speech_duration_ms: int = 250, max_speech_duration_s: float = float("inf"),
                          min_silence_duration_ms: int = 100, speech_pad_ms: int = 30, return_seconds: bool = False,
                          time_resolution: int = 1, visualize_probs: bool = False, progress_tracking_callback=None,
                          neg_threshold: float = None, window_size_samples: int = 512,
                          min_silence_at_max_speech: int = 98, use_max_poss_sil_at_max_speech: bool = True):
    """Single-clip reference signature (utils_vad.py:248-263); runs the whole clip in one device pass."""
    torch = _lib.require_gpu()
    if not torch.is_tensor(audio):
        try:
            audio = torch.Tensor(audio)
        except Exception:
            raise TypeError("Audio cannot be casted to tensor. Cast it manually")
    if len(audio.shape) > 1:
        for _ in range(len(audio.shape)):
            audio = audio.squeeze(0)
        if len(audio.shape) > 1:
            raise ValueError("More than one dimension in audio. Are you trying to process audio with 2 channels?")
    if visualize_probs:
        raise NotImplementedError("visualize_probs is outside the hot path (SURVEY §2 row 10)")
    res = get_speech_timestamps_batch(audio.unsqueeze(0), model, None, threshold, sampling_rate,
                                      min_speech_duration_ms, max_speech_duration_s, min_silence_duration_ms,
                                      speech_pad_ms, return_seconds, time_resolution, neg_threshold,
                                      min_silence_at_max_speech, use_max_poss_sil_at_max_speech)[0]
    if progress_tracking_callback:
        progress_tracking_callback(100.0)
    return res
