"""FireRedVAD / FireRedAED (non-stream) on MI355X: ORT-session boundary + the non-overlapping
window loop of FireRedVAD/Inference_FireRed_ONNX.py:523-613 (VAD) / :620-742 (AED, odim = 3)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import frontend as _frontend
from . import vadpost as _vadpost
from . import weights as _weights
from .fsmn import _Meta, pad_to_window_grid

SAMPLE_RATE, WINDOW_LENGTH, HOP_LENGTH = 16000, 400, 160


def valid_frame_count(num_samples):
    """snip_edges frame count (Inference_FireRed_ONNX.py:84-89, IN_SAMPLE_RATE == 16000)."""
    return 0 if num_samples < WINDOW_LENGTH else 1 + (num_samples - WINDOW_LENGTH) // HOP_LENGTH


class FireRedEngine:
    def __init__(self, weights=None, input_audio_length=16000, device="cuda:0"):
        torch = _lib.require_gpu()
        self.torch = torch
        self.device = torch.device(device)
        w = _weights.firered_synthetic(1234) if weights is None else weights
        c = dict(w["cfg"])
        w = {k: (np.ascontiguousarray(np.asarray(v), dtype=np.float32) if k != "cfg" else v) for k, v in w.items()}
        self.L = int(input_audio_length)
        self.fe = _frontend.Frontend("firered", self.L, device=device)
        self.T = self.fe.frames
        self.odim = c["odim"]
        cfg = _lib.FireRedCfg()
        for k in ("idim", "R", "M", "H", "P", "N1", "S1", "N2", "S2", "odim"):
            setattr(cfg, k, int(c[k]))
        cfg.frames = self.T
        self.cfg = cfg
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
        n = Lb.vadx_firered_packed_floats(C.byref(cfg))
        if n == 0:
            raise ValueError("FireRed config not supported by the HIP kernel (idim 80, H<=256, P<=128, frames<=112, odim<=4)")
        packed = np.zeros(n, dtype=np.float32)
        _lib.check(Lb.vadx_firered_pack_host(C.byref(cfg), C.byref(hw), packed.ctypes.data))
        self.packed = torch.from_numpy(packed).to(self.device)

    def run(self, audio_i16, windows_per_clip=1):
        """audio int16 [B, W*L] -> probs f32 [B*W, odim, T] (each window stateless, as the reference)."""
        t = self.torch
        if not t.is_tensor(audio_i16):
            audio_i16 = t.from_numpy(np.ascontiguousarray(audio_i16, dtype=np.int16))
        logmel = self.fe.logmel(audio_i16, windows_per_clip, self.L)
        nwin = logmel.shape[0]
        probs = t.empty((nwin, self.odim, self.T), dtype=t.float32, device=self.device)
        with t.cuda.device(self.device):
            _lib.check(_lib.lib().vadx_firered_run(C.byref(self.cfg), self.packed.data_ptr(), logmel.data_ptr(), nwin,
                                                   probs.data_ptr(), _lib.stream_ptr()))
        return probs

    def detect(self, clips_i16, pad_noise=None, post=(5, 0.4, 20, 2000, 20, 5, 0), return_probs=False):
        """Equal-length clips int16 [B,N] (host) -> per clip [(start_s, end_s)] for output channel 0
        (VAD driver :535-591).  pad_noise: standard-normal [B, >= pad] for the tail padding."""
        clips = np.asarray(clips_i16)
        B, n = clips.shape
        rows = [pad_to_window_grid(clips[b], self.L, self.L, None if pad_noise is None else pad_noise[b]) for b in range(B)]
        padded = np.stack(rows)
        W = padded.shape[1] // self.L
        probs = self.run(padded, W).view(B, W, self.odim, self.T)
        nvalid = valid_frame_count(n)
        track = probs[:, :, 0, :].reshape(B, W * self.T)[:, :nvalid].contiguous()
        pp = _vadpost.VadPostprocessor(*post, device=self.device)
        if nvalid == 0:
            out = [[] for _ in range(B)]
            return (out, track) if return_probs else out
        dec, segs, counts = pp.process_batch(track)
        segs, counts = segs.cpu().numpy(), counts.cpu().numpy()
        nfr = track.shape[1]     # min(valid frames, W*T): 10 s clips give 980 of the 998 snip-edge frames
        out = [pp.segments_to_seconds(segs[b, :counts[b]].tolist(), nfr, n / SAMPLE_RATE) for b in range(B)]
        return (out, track, dec) if return_probs else out


class FireRedSession:
    """onnxruntime.InferenceSession look-alike: {'audio': int16 [1,1,L]} -> [probs f32 [1,odim,T]]
    (FireRedVAD/Export_FireRedVAD.py:785-807); a leading batch of windows is accepted."""

    def __init__(self, weights=None, input_audio_length=16000, device="cuda:0"):
        self.engine = FireRedEngine(weights, input_audio_length, device)
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
