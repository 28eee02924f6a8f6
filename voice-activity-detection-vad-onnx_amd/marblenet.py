"""NVIDIA Frame-VAD MarbleNet v2.0 on MI355X: ORT-session boundary + the window loop of
NVIDIA_.../Inference_NVIDIA_MarbleNet_VAD_ONNX.py:130-147,369-402.

The encoder/decoder classes live in NeMo (not in the reference tree); weights come as the
checkpoint's conv + BatchNorm tensors and are BN-folded at load exactly like the reference's
`fold_bn_into_conv1d` (Export_NVIDIA_MarbleNet_VAD.py:58-105)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import checkpoints as _checkpoints
from . import frontend as _frontend
from . import vadpost as _vadpost
from . import weights as _weights
from .fsmn import _Meta, pad_to_window_grid

SAMPLE_RATE = 16000
OUTPUT_FRAME_SHIFT_S = 320 / 16000


def _pad16(a, axes):
    pads = [(0, 0)] * a.ndim
    for ax in axes:
        pads[ax] = (0, (-a.shape[ax]) % 16)
    return np.ascontiguousarray(np.pad(a, pads), dtype=np.float32)


class MarbleNetEngine:
    def __init__(self, weights=None, device="cuda:0", blocks=None, bn_eps=None, in_sample_rate=16000):
        """in_sample_rate: the export's IN_SAMPLE_RATE (Export_NVIDIA_MarbleNet_VAD.py:237-254): audio arrives at that rate and
        the graph resamples every window to 16 kHz itself."""
        torch = _lib.require_gpu()
        self.in_sample_rate = int(in_sample_rate)
        self.torch = torch
        self.device = torch.device(device)
        w = _checkpoints.resolve("marblenet", weights)
        blocks = _weights.MARBLENET_BLOCKS if blocks is None else blocks
        eps = _weights.MARBLENET_BN_EPS if bn_eps is None else bn_eps
        self.stages = []          # (cfg, dev tensors..., block_start flag)
        cin = 80
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)   # noqa: E731
        for bi, (filt, rep, k, stride, dil, residual, sep) in enumerate(blocks):
            block_cin = cin
            for r in range(rep):
                p = f"b{bi}r{r}"
                pw, pb = _weights.fold_bn(np.asarray(w[p + "_pw"])[:, :, None], None, w[p + "_gamma"], w[p + "_beta"],
                                          w[p + "_mean"], w[p + "_var"], eps)
                last = r == rep - 1
                cfg = _lib.SepConvCfg(cin, filt, k, stride, dil, 1 if sep else 0, block_cin if (residual and last) else 0, 1)
                st = {"cfg": cfg, "dw": dev(w[p + "_dw"]) if sep else None, "pw": dev(_lib.frag_major(pw[:, :, 0])),
                      "pb": dev(_pad16(pb, (0,))), "rw": None, "rb": None, "first": r == 0, "res": residual and last,
                      "pw_host": _pad16(pw[:, :, 0], (0, 1)), "rw_host": None}
                if residual and last:
                    rw, rb = _weights.fold_bn(np.asarray(w[f"b{bi}res_pw"])[:, :, None], None, w[f"b{bi}res_gamma"],
                                              w[f"b{bi}res_beta"], w[f"b{bi}res_mean"], w[f"b{bi}res_var"], eps)
                    st["rw"], st["rb"] = dev(_lib.frag_major(rw[:, :, 0])), dev(_pad16(rb, (0,)))
                    st["rw_host"] = _pad16(rw[:, :, 0], (0, 1))
                self.stages.append(st)
                cin = filt
        self.cout = cin
        self.dec_w, self.dec_b = dev(w["dec_w"]), dev(w["dec_b"])
        self._fe = {}
        # the published 3x2x64 layout runs on the fused kernels: one launch per residual block, one for blocks 5 + 6 + decoder
        self.fused = tuple(tuple(b) for b in blocks) == tuple(tuple(b) for b in _weights.MARBLENET_BLOCKS)
        # fp16 x 2 split products for the 1x1 convs of the fused residual blocks (csrc/split2.h; include/vadx.h: vadx_marblenet_cfg)
        self.arithmetic = None            # None = the module default (_lib.gemm_mode()); "h2" | "f32" ("split" has no form here: float32)
        self.h2_ok = False
        self.range_fallbacks = 0          # batches recomputed on float32 because an activation left the fp16 range
        self._flag = torch.zeros(2, dtype=torch.int32, device=self.device)
        if self.fused:
            self.h2_ok = True
            for k in range(0, 9):             # prologue + blocks 5 / 6 of the tail launch (plane order), blocks 2-4 (kgemm_h's k order)
                st = self.stages[k]
                for name in ("pw", "rw"):
                    host = st[name + "_host"]
                    # k order: the fused block whose input has 128 channels (stages 1, 2) splits its operand in the GEMM waves (K_QUARTER);
                    # the 64-channel blocks, the prologue and the tail read operand planes (K_PLAIN)
                    frags = None if host is None else _lib.frag_h2(host, _lib.H2_K_QUARTER if k in (1, 2) else _lib.H2_K_PLAIN)
                    st[name + "_h"] = None if frags is None else dev(frags)
                    self.h2_ok = self.h2_ok and (host is None or frags is not None)

    def mode(self):
        m = self.arithmetic or _lib.gemm_mode()
        return "h2" if (m == "h2" and self.h2_ok) else "f32"

    def h2_macs_per_out_frame(self):
        """Multiply-accumulates per encoder output frame that run as fp16 x 2 split products when mode() is "h2": every 1x1 conv whose
        `_h` fragments _run_fused hands to a launch (prologue, the fused blocks' point-wise + residual convs, both tail blocks).
        bench_models.flop_marblenet_h2_out_frame prices exactly this set on the fp16 pipe."""
        mac = 0
        for st in self.stages:
            for name in ("pw", "rw"):
                if st.get(name + "_h") is not None:
                    mac += (st["cfg"].cin if name == "pw" else st["cfg"].residual_cin) * st["cfg"].cout
        return mac

    def frontend(self, L):
        if L not in self._fe:
            self._fe[L] = _frontend.Frontend("marblenet", L, device=self.device, in_sample_rate=self.in_sample_rate)
        return self._fe[L]

    def run(self, audio_i16, windows_per_clip=1, window_len=None):
        """audio int16 [B, W*L] -> (score_silence, score_active f32 [B*W, T'], signal_len = T' - 1)."""
        t = self.torch
        if not t.is_tensor(audio_i16):
            audio_i16 = t.from_numpy(np.ascontiguousarray(audio_i16, dtype=np.int16))
        a = audio_i16.to(self.device)
        L = int(a.shape[-1] // windows_per_clip if window_len is None else window_len)
        fe = self.frontend(L)
        x = fe.logmel(a, windows_per_clip, L)               # [N, T, 80] time-major
        N, T = x.shape[0], x.shape[1]
        xs = (T * 80, 1, 80)
        cur, cur_T = x, T
        block_in = None
        lib = _lib.lib()
        if self.fused:
            mode = self.mode()
            out = self._run_fused(x, N, T, mode)
            if mode == "h2" and int(self._flag[0].item()) != 0:      # (synchronises) an activation left the fp16 range: float32 MFMAs
                self.range_fallbacks += 1
                self._flag.zero_()
                out = self._run_fused(x, N, T, "f32")
            return out
        with t.cuda.device(self.device):
            for st in self.stages:
                cfg = st["cfg"]
                if st["first"]:
                    block_in = cur
                pad = (cfg.dilation * (cfg.kernel - 1)) // 2
                t_out = (cur_T + 2 * pad - cfg.dilation * (cfg.kernel - 1) - 1) // cfg.stride + 1
                y = t.empty((N, cfg.cout, t_out), dtype=t.float32, device=self.device)
                _lib.check(lib.vadx_sepconv_block(C.byref(cfg), None if st["dw"] is None else st["dw"].data_ptr(),
                                                  st["pw"].data_ptr(), st["pb"].data_ptr(),
                                                  None if st["rw"] is None else st["rw"].data_ptr(),
                                                  None if st["rb"] is None else st["rb"].data_ptr(),
                                                  cur.data_ptr(), xs[0], xs[1], xs[2], cur_T,
                                                  block_in.data_ptr() if st["res"] else None, y.data_ptr(), N, t_out,
                                                  _lib.stream_ptr(), None))
                cur, cur_T = y, t_out
                xs = (cfg.cout * t_out, t_out, 1)
            s0 = t.empty((N, cur_T), dtype=t.float32, device=self.device)
            s1 = t.empty((N, cur_T), dtype=t.float32, device=self.device)
            _lib.check(lib.vadx_frame_classifier(cur.data_ptr(), self.dec_w.data_ptr(), self.dec_b.data_ptr(), N, self.cout,
                                                 cur_T, s0.data_ptr(), s1.data_ptr(), _lib.stream_ptr()))
        return s0, s1, cur_T - 1

    def run_from_host(self, host_i16, windows_per_clip=1, window_len=None, chunk_clips=256, feed=None):
        """`run` fed from HOST memory (int16 [B, W*L], ideally pinned: vadx.feed.pin), the upload of chunk k + 1 overlapped with the
        launches of chunk k (vadx.feed.HostPcmFeed); bit-identical to `run` of the resident batch."""
        from . import feed as _feed
        f = feed or _feed.HostPcmFeed(self.device, host_i16.shape[1], chunk_clips)
        return _feed.cat_results(f.map([host_i16], lambda a: self.run(a, windows_per_clip, window_len)))

    def _run_fused(self, x, N, T, mode="f32"):
        """Published layout: block 1 (stride 2, time-major log-mel in) through the per-sub-block entry, blocks 2-4 as one
        fused launch each (the tensor between a block's two sub-blocks stays in LDS), blocks 5 + 6 + Linear + softmax as one
        launch (nothing but the two scores per frame is written)."""
        t, lib, st = self.torch, _lib.lib(), self.stages
        p = lambda a: None if a is None else a.data_ptr()        # noqa: E731
        with t.cuda.device(self.device):
            s0 = st[0]
            cfg = s0["cfg"]
            pad = (cfg.dilation * (cfg.kernel - 1)) // 2
            T1 = (T + 2 * pad - cfg.dilation * (cfg.kernel - 1) - 1) // cfg.stride + 1
            cur = t.empty((N, cfg.cout, T1), dtype=t.float32, device=self.device)
            mc = _lib.MarbleNetCfg(_lib.ARITH[mode], 0, self._flag.data_ptr())
            sfx = "_h" if mode == "h2" else ""
            _lib.check(lib.vadx_sepconv_block(C.byref(cfg), p(s0["dw"]), p(s0["pw" + sfx]), p(s0["pb"]), None, None, x.data_ptr(),
                                              T * 80, 1, 80, T, None, cur.data_ptr(), N, T1, _lib.stream_ptr(), C.byref(mc)))
            for k in (1, 3, 5):
                a, b = st[k], st[k + 1]
                y = t.empty((N, 64, T1), dtype=t.float32, device=self.device)
                _lib.check(lib.vadx_marblenet_block2(a["cfg"].cin, a["cfg"].kernel, p(a["dw"]), p(a["pw" + sfx]), p(a["pb"]), p(b["dw"]),
                                                     p(b["pw" + sfx]), p(b["pb"]), p(b["rw" + sfx]), p(b["rb"]), cur.data_ptr(), y.data_ptr(),
                                                     N, T1, _lib.stream_ptr(), C.byref(mc)))
                cur = y
            s0o = t.empty((N, T1), dtype=t.float32, device=self.device)
            s1o = t.empty((N, T1), dtype=t.float32, device=self.device)
            _lib.check(lib.vadx_marblenet_tail(p(st[7]["dw"]), p(st[7]["pw" + sfx]), p(st[7]["pb"]), p(st[8]["pw" + sfx]), p(st[8]["pb"]),
                                               self.dec_w.data_ptr(), self.dec_b.data_ptr(), cur.data_ptr(), s0o.data_ptr(),
                                               s1o.data_ptr(), N, T1, _lib.stream_ptr(), C.byref(mc)))
        return s0o, s1o, T1 - 1

    def detect(self, clips_i16, window_len=None, pad_noise=None, post=(3, 0.5, 10, 1000, 10, 3, 0), return_probs=False):
        """Equal-length clips int16 [B,N] (host) -> per clip [(start_s, end_s)].  window_len None = the
        dynamic-axis mode (one window = the whole clip, up to 3600 s, :130-135)."""
        clips = np.asarray(clips_i16)
        B, n = clips.shape
        L = min(self.in_sample_rate * 3600, n) if window_len is None else int(window_len)
        rows = [pad_to_window_grid(clips[b], L, L, None if pad_noise is None else pad_noise[b]) for b in range(B)]
        padded = np.stack(rows)
        W = padded.shape[1] // L
        s0, s1, slen = self.run(padded, W, L)
        valid = min(slen, s1.shape[1])
        track = s1.view(B, W, -1)[:, :, :valid].reshape(B, W * valid).contiguous()
        pp = _vadpost.VadPostprocessor(*post, frame_shift_s=OUTPUT_FRAME_SHIFT_S, frame_length_s=None, device=self.device)
        dec, segs, counts = pp.process_batch(track)
        segs, counts = segs.cpu().numpy(), counts.cpu().numpy()
        out = [pp.segments_to_seconds(segs[b, :counts[b]].tolist(), track.shape[1], n / self.in_sample_rate) for b in range(B)]
        return (out, track, dec) if return_probs else out


class MarbleNetSession:
    """{'audio': int16 [1,1,L]} -> [score_silence [1,T,1], score_active [1,T,1], signal_len int32 [1]]
    (Export_NVIDIA_MarbleNet_VAD.py:436-457; dynamic audio length).  io_dtype="float16": I/O-compatible with the reference's
    fp16-optimised model (Optimize_ONNX.py:37-44 converts with keep_io_types=False, so the two scores leave as float16) -- the
    arithmetic here stays float32 and the scores are rounded once on the way out."""

    def __init__(self, weights=None, device="cuda:0", in_sample_rate=16000, io_dtype="float32"):
        if io_dtype not in ("float32", "float16"):
            raise ValueError("io_dtype must be 'float32' or 'float16'")
        self.io_dtype = np.float16 if io_dtype == "float16" else np.float32
        ftype = "tensor(float16)" if io_dtype == "float16" else "tensor(float)"
        self.engine = MarbleNetEngine(weights, device, in_sample_rate=in_sample_rate)
        self._inputs_meta = [_Meta("audio", [1, 1, "audio_len"], "tensor(int16)")]
        self._outputs_meta = [_Meta("score_silence", [1, "signal_len", 1], ftype),
                              _Meta("score_active", [1, "signal_len", 1], ftype),
                              _Meta("signal_len", [1], "tensor(int32)")]

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
        s0, s1, slen = self.engine.run(audio.reshape(-1, audio.shape[-1]))
        res = {"score_silence": s0.cpu().numpy()[:, :, None].astype(self.io_dtype), "score_active": s1.cpu().numpy()[:, :, None].astype(self.io_dtype),
               "signal_len": np.array([slen], dtype=np.int32)}
        names = ["score_silence", "score_active", "signal_len"] if output_names is None else output_names
        return [res[n] for n in names]
