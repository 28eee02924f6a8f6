"""ctypes binding of libvadx.so (the C ABI declared in include/vadx.h).

The product path has NO CPU fallback: if the HIP library is missing or a GPU call fails this
module raises -- it never silently routes anywhere else.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VADX_LIBRARY") or os.path.join(HERE, "libvadx.so")     # override: a differently built libvadx
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "vadx.h")


def _header_abi_version():
    """`#define VADX_ABI_VERSION n` of include/vadx.h -- the one place the ABI number is written.  A deployment that ships only this
    package and libvadx.so has no header: build.py then left the number it compiled against in _abi.py next to the library."""
    import re
    if os.path.exists(HEADER_PATH):
        with open(HEADER_PATH) as fh:
            m = re.search(r"^#define\s+VADX_ABI_VERSION\s+(\d+)\s*$", fh.read(), flags=re.M)
        if not m:
            raise RuntimeError(f"{HEADER_PATH}: no '#define VADX_ABI_VERSION <n>' line")
        return int(m.group(1))
    try:
        from ._abi import ABI_VERSION as built       # written by build.py
        return int(built)
    except ImportError as e:
        raise RuntimeError(f"neither {HEADER_PATH} nor {os.path.join(HERE, '_abi.py')} exists: cannot tell which ABI this binding expects") from e


ABI_VERSION = _header_abi_version()


class VadxError(RuntimeError):
    pass


class SileroWeightsHost(C.Structure):
    _fields_ = [("stft_basis", C.c_void_p), ("enc_w", C.c_void_p * 4), ("enc_b", C.c_void_p * 4),
                ("lstm_w_ih", C.c_void_p), ("lstm_w_hh", C.c_void_p), ("lstm_b_ih", C.c_void_p),
                ("lstm_b_hh", C.c_void_p), ("dec_w", C.c_void_p), ("dec_b", C.c_void_p)]


class MarbleNetCfg(C.Structure):
    """vadx_marblenet_cfg (include/vadx.h): arithmetic of the fused MarbleNet block launches + the fp16 x 2 range flag words"""
    _fields_ = [("arithmetic", C.c_int32), ("reserved", C.c_int32), ("range_flag", C.c_void_p)]


class SileroCfg(C.Structure):
    """vadx_silero_cfg (include/vadx.h): per-call configuration of the Silero launches"""
    _fields_ = [("arithmetic", C.c_int32), ("reserved", C.c_int32 * 3)]


ARITH = {"auto": 0, "f32": 1, "split": 2, "bf16x3": 2, "h2": 3, "f16x2": 3}       # VADX_ARITH_*


class SileroSegParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("neg_threshold", C.c_double), ("sampling_rate", C.c_int),
                ("min_speech_duration_ms", C.c_double), ("max_speech_duration_s", C.c_double),
                ("min_silence_duration_ms", C.c_double), ("speech_pad_ms", C.c_double),
                ("min_silence_at_max_speech", C.c_double), ("use_max_poss_sil_at_max_speech", C.c_int)]


class FrontendCfg(C.Structure):
    _fields_ = [("prep", C.c_int), ("k0", C.c_float), ("k1", C.c_float), ("center_pad", C.c_int),
                ("tap0", C.c_int), ("taps", C.c_int), ("hop", C.c_int), ("n_bins", C.c_int),
                ("n_mels", C.c_int), ("log_mode", C.c_int), ("log_floor", C.c_float), ("frames", C.c_int),
                ("window_len", C.c_int), ("in_window_len", C.c_int), ("rs_scale", C.c_float), ("fold", C.c_int)]


class FsmnDims(C.Structure):
    _fields_ = [("input_affine_dim", C.c_int), ("linear_dim", C.c_int), ("output_affine_dim", C.c_int),
                ("output_dim", C.c_int), ("frames", C.c_int), ("speech_2_noise_ratio", C.c_float), ("arithmetic", C.c_int)]


class FsmnWeightsHost(C.Structure):
    _fields_ = [("in1_w", C.c_void_p), ("in1_b", C.c_void_p), ("in2_w", C.c_void_p), ("in2_b", C.c_void_p),
                ("lin_w", C.c_void_p * 4), ("fir_w", C.c_void_p * 4), ("aff_w", C.c_void_p * 4),
                ("aff_b", C.c_void_p * 4), ("out1_w", C.c_void_p), ("out1_b", C.c_void_p),
                ("out2_w", C.c_void_p), ("out2_b", C.c_void_p), ("cmvn_means", C.c_void_p),
                ("cmvn_vars", C.c_void_p)]


class FsmnLoopParams(C.Structure):
    _fields_ = [("look_backward", C.c_int), ("one_minus_speech_threshold", C.c_float),
                ("noise_db_init", C.c_float), ("snr_threshold", C.c_float), ("speaking_score", C.c_double),
                ("silence_score", C.c_double)]


class FireRedCfg(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("idim", "R", "M", "H", "P", "N1", "S1", "N2", "S2", "odim", "frames", "arithmetic")]


class FireRedWeightsHost(C.Structure):
    _fields_ = [("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p),
                ("fsmn_lb", C.c_void_p * 16), ("fsmn_la", C.c_void_p * 16), ("blk_fc1_w", C.c_void_p * 16),
                ("blk_fc1_b", C.c_void_p * 16), ("blk_fc2_w", C.c_void_p * 16), ("dnn_w", C.c_void_p * 4),
                ("dnn_b", C.c_void_p * 4), ("out_w", C.c_void_p), ("out_b", C.c_void_p)]


class VadPostParams(C.Structure):
    _fields_ = [("smooth_window_size", C.c_int), ("prob_threshold", C.c_float), ("min_speech_frame", C.c_int),
                ("max_speech_frame", C.c_int), ("min_silence_frame", C.c_int), ("merge_silence_frame", C.c_int),
                ("extend_speech_frame", C.c_int)]


class StreamVadPostParams(C.Structure):
    _fields_ = [("smooth_window_size", C.c_int), ("speech_threshold", C.c_float), ("pad_start_frame", C.c_int),
                ("min_speech_frame", C.c_int), ("max_speech_frame", C.c_int), ("min_silence_frame", C.c_int)]


class SepConvCfg(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("cin", "cout", "kernel", "stride", "dilation", "depthwise", "residual_cin", "relu")]


class FtView(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("c_total", C.c_int), ("c_off", C.c_int), ("c", C.c_int)]


class FtLn(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p)]


class DfsmnCfbWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("ln0_w", "gate_w", "in_w", "in_b", "front_tab", "conv_w", "fwd_tbl", "fwd_fix", "lin_w", "lin_b",
                                          "inv_tbl", "out_fix", "fwd_tbl_q", "inv_tbl_q")] + [("front_arithmetic", C.c_int32), ("back_arithmetic", C.c_int32)]


class DfsmnMaskWeights(C.Structure):
    _fields_ = [("hidden", C.c_int), ("fsmn_hidden", C.c_int), ("layers", C.c_int), ("lorder", C.c_int),
                ("shift", C.c_void_p), ("scale", C.c_void_p), ("linear1_w", C.c_void_p), ("linear1_b", C.c_void_p),
                ("linear3_w", C.c_void_p), ("linear3_b", C.c_void_p), ("fsmn_linear_w", C.c_void_p * 8),
                ("fsmn_linear_b", C.c_void_p * 8), ("fsmn_project_w", C.c_void_p * 8), ("fsmn_conv_w", C.c_void_p * 8)]


_P, _I, _L, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_size_t

# name -> (restype, argtypes); every symbol include/vadx.h declares
SIGNATURES = {
    "vadx_abi_version": (_I, []),
    "vadx_last_error": (C.c_char_p, []),
    "vadx_silero_packed_floats": (_Z, []),
    "vadx_silero_pack_host": (_I, [C.POINTER(SileroWeightsHost), _P]),
    "vadx_silero_workspace_bytes": (_Z, [_I, _I]),
    # (every Silero launch ends with (stream, const vadx_silero_cfg *); None = the library's defaults)
    "vadx_silero_step": (_I, [_P, _P, _P, _L, _I, _P, _P, _P, _Z, _P, _P]),
    "vadx_silero_clips": (_I, [_P, _P, _I, _L, _L, _P, _P, _P, _Z, _P, _P]),
    "vadx_silero_encode": (_I, [_P, _P, _I, _L, _L, _P, _Z, _P, _P]),
    "vadx_silero_encode_pcm16": (_I, [_P, _P, C.c_float, _I, _L, _L, _P, _Z, _P, _P]),
    "vadx_silero_encode_pcm16_part": (_I, [_P, _P, C.c_float, _I, _L, _L, _I, _I, _P, _Z, _P, _P]),
    "vadx_silero_recur": (_I, [_P, _P, _Z, _I, _I, _P, _P, _P, _P, _P]),
    "vadx_silero_encode_span": (_I, [_P, _P, _I, _L, _L, _I, _I, _P, _Z, _P, _P]),
    "vadx_silero_recur_span": (_I, [_P, _P, _Z, _I, _I, _P, _P, _L, _P, _P, _P]),
    "vadx_silero_segments": (_I, [_P, _I, _I, _P, C.POINTER(SileroSegParams), _P, _P, _I, _P]),
    "vadx_silero_range_flag": (_I, [_P, _I, _P, _P, _P]),
    "vadx_frontend_packed_floats": (_Z, [C.POINTER(FrontendCfg)]),
    "vadx_frontend_pack_host": (_I, [C.POINTER(FrontendCfg), _P, _P, _I, _P, _P, _P]),
    "vadx_frontend_fold_kind": (_I, [C.POINTER(FrontendCfg), _P, _P, _I]),
    "vadx_frontend_logmel": (_I, [C.POINTER(FrontendCfg), _P, _P, _P, _L, _L, _I, _I, _P, _P, _P]),
    "vadx_fsmn_packed_floats": (_Z, [C.POINTER(FsmnDims)]),
    "vadx_fsmn_range_flag": (_I, [C.POINTER(FsmnDims), _P, _I, _P, _P, _P]),
    "vadx_fsmn_pack_host": (_I, [C.POINTER(FsmnDims), C.POINTER(FsmnWeightsHost), _P]),
    "vadx_fsmn_energy": (_I, [_P, _L, _L, _I, _I, _I, _I, _P, _P, _P]),
    "vadx_fsmn_window_stats": (_I, [_P, _L, _L, _I, _I, _I, _I, _P, _P, _P]),
    "vadx_frontend_logmel_means": (_I, [C.POINTER(FrontendCfg), _P, _P, _P, _L, _L, _I, _I, _P, _P, _P]),
    "vadx_frontend_window_means": (_I, [_P, _L, _L, _I, _I, _I, C.c_float, _P, _P]),
    "vadx_fsmn_run": (_I, [C.POINTER(FsmnDims), _P, _P, _P, C.POINTER(C.c_void_p * 4), C.POINTER(C.c_void_p * 4),
                           _P, _P, _I, _P, _P, _P, _P]),
    "vadx_fsmn_clips": (_I, [C.POINTER(FsmnDims), _P, _P, _P, _I, _I, C.POINTER(FsmnLoopParams), _P, _P, _P, _P]),
    "vadx_firered_packed_floats": (_Z, [C.POINTER(FireRedCfg)]),
    "vadx_firered_range_flag": (_I, [C.POINTER(FireRedCfg), _P, _I, _P, _P, _P]),
    "vadx_firered_pack_host": (_I, [C.POINTER(FireRedCfg), C.POINTER(FireRedWeightsHost), _P]),
    "vadx_firered_run": (_I, [C.POINTER(FireRedCfg), _P, _P, _I, _P, _P]),
    "vadx_ingest_out_frames": (C.c_int64, [C.c_int64, _I, _I]),
    "vadx_ingest_pcm16": (_I, [_P, C.c_int64, _I, C.c_int64, _I, _I, _P, C.c_int64, _I, _P]),
    "vadx_frag_major_floats": (C.c_size_t, [_I, _I]),
    "vadx_frag_major_host": (_I, [_P, _I, _I, _P]),
    "vadx_firered_stream_run": (_I, [C.POINTER(FireRedCfg), _P, _P, _I, _P, _P, _P, _P]),
    "vadx_stream_vadpost_state_bytes": (_Z, [_I]),
    "vadx_stream_vadpost": (_I, [C.POINTER(StreamVadPostParams), _P, _L, _I, _I, _P, _I, _I, _P, _P, _I, _P]),
    "vadx_vadpost_workspace_bytes": (_Z, [_I, _I]),
    "vadx_vadpost": (_I, [C.POINTER(VadPostParams), _P, _I, _P, _I, _P, _P, _P, _I, _P, _Z, _P]),
    "vadx_sepconv_block": (_I, [C.POINTER(SepConvCfg), _P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _I, _I, _P, C.POINTER(MarbleNetCfg)]),
    "vadx_marblenet_block2": (_I, [_I, _I] + [_P] * 10 + [_I, _I, _P, C.POINTER(MarbleNetCfg)]),
    "vadx_frag_h2_floats": (C.c_size_t, [_I, _I]),
    "vadx_frag_h2_host": (_I, [_P, _I, _I, _I, _P, _P]),
    "vadx_marblenet_tail": (_I, [_P] * 10 + [_I, _I, _P, C.POINTER(MarbleNetCfg)]),
    "vadx_frame_classifier": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "vadx_dfsmn_frame_stats": (_I, [C.POINTER(FtView), C.POINTER(FtView), _I, _I, _P, _P]),
    "vadx_dfsmn_stats_merge": (_I, [_P, _P, _I, _P, _P]),
    "vadx_dfsmn_pw_conv": (_I, [_I, C.POINTER(FtView), C.POINTER(FtView), C.POINTER(FtLn), _P, _P, _P, _P,
                                C.POINTER(FtView), C.POINTER(FtView), C.POINTER(FtView), _I, _I, _I, _I, _I, _P, _P, _P]),
    "vadx_dfsmn_dft_f": (_I, [_I, C.POINTER(FtView), C.POINTER(FtView), C.POINTER(FtLn), _P, C.POINTER(FtView), _I, _I, _P, _P]),
    "vadx_dfsmn_lstm_f": (_I, [C.POINTER(FtView), C.POINTER(FtLn), C.POINTER(C.c_void_p * 2), C.POINTER(C.c_void_p * 2),
                               C.POINTER(C.c_void_p * 2), C.POINTER(C.c_void_p * 2), C.POINTER(FtView), _I, _I, _P, _I, _P]),
    "vadx_dfsmn_cfb_front": (_I, [C.POINTER(DfsmnCfbWeights), C.POINTER(FtView), C.POINTER(FtView), _P, _P, _P, _P, _P, _I, _P]),
    "vadx_dfsmn_cfb_back": (_I, [C.POINTER(DfsmnCfbWeights), _P, _P, _P, _P, C.POINTER(FtView), _P, _I, _P]),
    "vadx_dfsmn_alpha_scale": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "vadx_dfsmn_lstm_t": (_I, [_I, C.POINTER(FtView), C.POINTER(FtLn), C.POINTER(C.c_void_p * 2), C.POINTER(C.c_void_p * 2),
                               C.POINTER(C.c_void_p * 2), C.POINTER(C.c_void_p * 2), _P, _P, C.POINTER(FtView),
                               C.POINTER(FtView), _I, _I, _I, _P]),
    "vadx_dfsmn_lstm_t_ex": (_I, [_I, C.POINTER(FtView), C.POINTER(FtLn), C.POINTER(C.c_void_p * 2), C.POINTER(C.c_void_p * 2),
                                  C.POINTER(C.c_void_p * 2), C.POINTER(C.c_void_p * 2), _P, _P, C.POINTER(FtView),
                                  C.POINTER(FtView), _I, _I, _I, _I, _P, _I, _P]),
    "vadx_dfsmn_ft_repack": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vadx_dfsmn_istft": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "vadx_dfsmn_vote": (_I, [_P, _I, _I, _I, _I, C.c_double, C.c_double, _P, _P]),
    "vadx_frontend_logmel_ex": (_I, [C.POINTER(FrontendCfg), _P, _P, _P, _L, _L, _I, _I, _P, _P, _I, _I, _P, _P]),
    "vadx_frontend_stft_ft": (_I, [C.POINTER(FrontendCfg), _P, _P, _L, _L, _I, _I, _P, _P, _I, _I, _P]),
    "vadx_dfsmn_mask_net": (_I, [C.POINTER(DfsmnMaskWeights), _P, _I, _I, _P, _P]),
}

_lib = None
_trace = None          # list of (entry name, start event, stop event) while a `trace()` block is open


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VadxError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch first: it ships its own libamdhip64.  If libvadx.so were loaded before torch, it would bind the system HIP runtime
        # and the process would hold two runtimes -- the second one then finds "no ROCm-capable device" (seen when build() and
        # smoke() ran in one process).  With torch's runtime already mapped, the loader resolves libvadx.so against that same copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        got = handle.vadx_abi_version()
        if got != ABI_VERSION:
            raise VadxError(f"{LIB_PATH} was built as ABI {got}, include/vadx.h says {ABI_VERSION}: rebuild it "
                            "(python -m vadx.build --force)")
        _lib = handle
    return _lib


class _TracingLib:
    """Same attributes as the CDLL handle; every stream-ordered entry point is bracketed by HIP events on the
    stream it is launched on (torch's current stream = the stream handed to the C ABI)."""

    def __getattr__(self, name):
        fn = getattr(_load(), name)
        if name.endswith(("_host", "_floats", "_bytes", "_version", "_error", "_frames")):
            return fn
        import torch

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            if _trace is not None:
                _trace.append((name, e0, e1))
            return rc
        return timed


_tracing = _TracingLib()


def lib():
    """Load libvadx.so (built in-tree by `python -m vadx.build` / __graft_entry__.build())."""
    return _tracing if _trace is not None else _load()


class trace:
    """`with _lib.trace() as tr: engine.run(...)` -> tr.ms = {entry point: total device ms}, tr.calls = {entry: n}
    (HIP events around each C-ABI launch; bench.py's per-kernel roofline numbers come from here)."""

    def __enter__(self):
        global _trace
        _trace = []
        self.ms, self.calls = {}, {}
        return self

    def __exit__(self, *exc):
        global _trace
        rec, _trace = _trace, None
        import torch
        torch.cuda.synchronize()
        for name, e0, e1 in rec:
            self.ms[name] = self.ms.get(name, 0.0) + e0.elapsed_time(e1)
            self.calls[name] = self.calls.get(name, 0) + 1
        return False


GEMM_MODES = {"f32": ARITH["f32"], "split": ARITH["split"], "h2": ARITH["h2"]}      # name -> VADX_ARITH_*
_gemm_default = [None]


def gemm_mode(mode=None):
    """The arithmetic FSMN / FireRed / DFSMN engines use for their dense layers unless told otherwise per engine (`engine.arithmetic`):
    "f32" MFMAs, "split" = bf16 x 3 exact-split products, "h2" = fp16 x 2 split products (the default; activations outside the fp16
    range are detected and the batch recomputed on "split").  A Python-side default only: the C ABI takes the arithmetic with every
    cfg / dims struct (include/vadx.h: VADX_ARITH_*).  Returns the mode that was active; None only queries.  Initial value: VADX_GEMM,
    read once here."""
    if _gemm_default[0] is None:
        e = os.environ.get("VADX_GEMM", "").strip().lower()
        _gemm_default[0] = {"0": "f32", "1": "split", "2": "h2", "bf16x3": "split", "f16x2": "h2"}.get(e, e) if e else "h2"
        if _gemm_default[0] not in GEMM_MODES:
            raise ValueError(f"VADX_GEMM must be one of {sorted(GEMM_MODES)}, got {e!r}")
    prev = _gemm_default[0]
    if mode is not None:
        if mode not in GEMM_MODES:
            raise ValueError(f"gemm mode must be one of {sorted(GEMM_MODES)}, got {mode!r}")
        _gemm_default[0] = mode
    return prev


class ArithBlobs:
    """What an engine whose packed blob depends on the arithmetic keeps: one (cfg struct, device blob) per arithmetic, built on first use
    by `build(mode) -> (cfg, device tensor)` (raising ValueError when the weights cannot be packed for that mode), and the fp16 x 2
    range protocol around a launch: `guarded(run)` runs `run(mode, cfg, blob)`; on "h2" the result stands only if the kernels' sticky
    range flag stayed clear (`flag(cfg, blob) -> (flag, amax)`, which synchronises), otherwise the batch is recomputed on "split"."""

    def __init__(self, build, flag):
        self._build, self._flag = build, flag
        self._blobs = {}
        self.arithmetic = None            # None = the module default (gemm_mode()); or "f32" | "split" | "h2"
        self.h2_ok = True
        self.range_fallbacks = 0

    def mode(self):
        m = self.arithmetic or gemm_mode()
        if m == "h2" and self.h2_ok and "h2" not in self._blobs:
            try:
                self._blobs["h2"] = self._build("h2")
            except ValueError:                # a weight outside the fp16 range (pack_host says so): this engine runs bf16 x 3
                self.h2_ok = False
        return "split" if (m == "h2" and not self.h2_ok) else m

    def get(self, mode=None):
        mode = mode or self.mode()
        if mode not in self._blobs:
            self._blobs[mode] = self._build(mode)
        return self._blobs[mode]

    def guarded(self, run):
        m = self.mode()
        out = run(m, *self.get(m))
        if m == "h2":
            flag, _ = self._flag(*self.get(m))
            if flag:
                self.range_fallbacks += 1
                out = run("split", *self.get("split"))
        return out


def check(rc, exc=VadxError):
    if rc != 0:
        msg = lib().vadx_last_error().decode("utf-8", "replace")
        raise (ValueError if rc == -1 else exc)(msg or f"libvadx error {rc}")


def frag_major(a):
    """row-major float32 [rows][cols] -> the fragment-major buffer (1-D, zero-padded to multiples of 16)
    the GEMM kernels stream their weights from (include/vadx.h: vadx_frag_major_host)."""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("frag_major expects a 2-D matrix")
    out = np.empty(lib().vadx_frag_major_floats(a.shape[0], a.shape[1]), dtype=np.float32)
    check(lib().vadx_frag_major_host(a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data))
    return out


H2_K_PLAIN, H2_K_QUARTER = 0, 1       # include/vadx.h: VADX_H2_K_*


def frag_h2(a, k_order=H2_K_PLAIN):
    """row-major float32 [rows][cols] -> fp16 x 2 fragments (include/vadx.h: vadx_frag_h2_host) as a float32-typed 1-D buffer, or None when
    a weight lies outside the fp16 range (the caller keeps that matrix on float32)."""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("frag_h2 expects a 2-D matrix")
    out = np.empty(lib().vadx_frag_h2_floats(a.shape[0], a.shape[1]), dtype=np.float32)
    if lib().vadx_frag_h2_host(a.ctypes.data, a.shape[0], a.shape[1], int(k_order), out.ctypes.data, None) != 0:
        return None
    return out


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise VadxError("no HIP device visible: the vadx product path runs only on an MI355X (no CPU fallback)")
    return torch


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def row_stride(t):
    """Element stride between rows of a contiguous [B, N] tensor (size-1 dims may report any stride)."""
    return int(t.stride(0)) if t.shape[0] > 1 and t.stride(0) >= t.shape[1] else int(t.shape[1])
