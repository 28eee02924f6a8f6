"""DFSMN near+far VAD on MI355X (reference: DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py,
Inference_DFSMN_VAD_ONNX.py).  This module is plumbing: it owns the device weights/tables and strings
the libvadx kernels together in the order of `DFSMN_VAD.forward` / `NET.forward`; all arithmetic is HIP.

Activations use the frame-tiled "FT" layout documented in csrc/dfsmn.hip."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

F_BINS, CEPS_F, CH = 160, 81, 20


def ft_tiles(frames):
    return (frames + 15) // 16


class FT:
    """A device tensor in FT layout + helpers to make channel-slice views."""

    def __init__(self, torch, device, n_chunks, frames, channels, bins):
        self.nt = ft_tiles(frames)
        self.tiles = n_chunks * self.nt
        self.C, self.F, self.frames = channels, bins, frames
        self.data = torch.zeros((self.tiles, channels, bins, 16), dtype=torch.float32, device=device)

    def view(self, c_off=0, c=None):
        return _lib.FtView(self.data.data_ptr(), self.C, c_off, self.C - c_off if c is None else c)


def to_ft(torch, x, device):
    """[N, C, F, T] (host or device) -> FT tensor object."""
    N, Cc, Fb, T = x.shape
    ft = FT(torch, device, N, T, Cc, Fb)
    pad = ft.nt * 16 - T
    xp = torch.nn.functional.pad(x.to(device, torch.float32), (0, pad))
    ft.data.copy_(xp.view(N, Cc, Fb, ft.nt, 16).permute(0, 3, 1, 2, 4).reshape(ft.tiles, Cc, Fb, 16))
    return ft


def from_ft(ft, n_chunks):
    """FT tensor -> [N, C, F, T] torch tensor on the device."""
    d = ft.data.view(n_chunks, ft.nt, ft.C, ft.F, 16).permute(0, 2, 3, 1, 4).reshape(n_chunks, ft.C, ft.F, ft.nt * 16)
    return d[..., :ft.frames]


def _pad_rows16(a):
    return np.ascontiguousarray(np.pad(a, ((0, (-a.shape[0]) % 16), (0, 0))), dtype=np.float32)


class Iccrn:
    """Device-side SDAEC ICCRN (`NET`): weights repacked for the pw_conv / dft_f / lstm kernels."""

    def __init__(self, weights, device="cuda:0"):
        self.torch = t = _lib.require_gpu()
        self.device = t.device(device)
        self.lib = _lib.lib()
        w = {k[len("iccrn."):]: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in weights.items()
             if k.startswith("iccrn.")}
        self.w = w
        self.d = {}

        def dev(name, arr):
            self.d[name] = t.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)

        for k, v in w.items():
            if k.endswith("conv_gate.weight") or k.endswith("conv_input.weight") or k in ("in_conv.weight", "out_conv.weight"):
                dev(k, _pad_rows16(v[:, :, 0, 0]))                                   # [co_pad][cin]
            elif k.endswith(".conv.weight"):                                         # (3,1) conv: [co][ci][3][1] -> [co_pad][3*ci]
                dev(k, _pad_rows16(np.transpose(v[:, :, :, 0], (0, 2, 1)).reshape(v.shape[0], -1)))
            elif k.endswith("linear.weight"):
                dev(k, _pad_rows16(v))
            elif k.endswith(".w") or k.endswith(".b"):                               # LayerNorm [1,C,F,1] -> [C][F]
                dev(k, v.reshape(v.shape[1], v.shape[2]))
            elif k.endswith("bias"):
                dev(k, np.pad(v, (0, (-v.shape[0]) % 16)))
            else:
                dev(k, v)
        # CepsUnit tables (Export_DFSMN_VAD.py:104-130), float32 torch ops in the reference's order
        n, half = 160, 80
        tt = t.arange(n, dtype=t.float32).unsqueeze(0)
        ff = t.arange(half + 1, dtype=t.float32).unsqueeze(1)
        omega = 2 * t.pi * ff * tt / n
        cos_k, sin_k = t.cos(omega), -t.sin(omega)                                   # [81,160], window = ones
        fwd = t.zeros((192, 160), dtype=t.float32)
        fwd[:81], fwd[96:96 + 81] = cos_k, sin_k
        fb = t.fft.fft(t.eye(n, dtype=t.float32))
        basis = t.vstack([t.real(fb[:half + 1]), t.imag(fb[:half + 1])]).float()
        inv_basis = t.linalg.pinv(basis).T                                            # [162,160]
        inv = t.zeros((160, 164), dtype=t.float32)
        inv[:, :162] = inv_basis.t()
        self.tbl_fwd, self.tbl_inv = fwd.to(self.device), inv.to(self.device)

    # ---- small helpers around the C ABI -------------------------------------------------------
    def _p(self, name):
        return self.d[name].data_ptr()

    def _ln(self, stats, prefix):
        return _lib.FtLn(stats.data_ptr(), self._p(prefix + ".w"), self._p(prefix + ".b"))

    def stats(self, a, b, F, tiles):
        t = self.torch
        s = t.empty((tiles, 16, 2), dtype=t.float32, device=self.device)
        _lib.check(self.lib.vadx_dfsmn_frame_stats(C.byref(a), None if b is None else C.byref(b), F, tiles, s.data_ptr(),
                                                   _lib.stream_ptr()))
        return s

    def pw(self, mode, a, b, ln, w, bias, out0, F, co, kf=1, act=0, w2=None, bias2=None, add=None, out1=None, tiles=None):
        _lib.check(self.lib.vadx_dfsmn_pw_conv(mode, C.byref(a), None if b is None else C.byref(b),
                                               None if ln is None else C.byref(ln), self._p(w), self._p(bias),
                                               None if w2 is None else self._p(w2), None if bias2 is None else self._p(bias2),
                                               None if add is None else C.byref(add), C.byref(out0),
                                               None if out1 is None else C.byref(out1), F, co, kf, act, tiles,
                                               _lib.stream_ptr()))

    def lstm_f(self, prefix, inp, ln, out, F, tiles):
        arr = lambda n: (C.c_void_p * 2)(self._p(f"{prefix}.lstm2.{n}_l0"), self._p(f"{prefix}.lstm2.{n}_l0_reverse"))   # noqa: E731
        wi, wh, bi, bh = arr("weight_ih"), arr("weight_hh"), arr("bias_ih"), arr("bias_hh")
        _lib.check(self.lib.vadx_dfsmn_lstm_f(C.byref(inp), None if ln is None else C.byref(ln), C.byref(wi), C.byref(wh),
                                              C.byref(bi), C.byref(bh), C.byref(out), F, tiles, _lib.stream_ptr()))

    # ---- CFB (:76-93) ---------------------------------------------------------------------------
    def cfb(self, name, a, b, out, n_chunks, frames, scratch=None):
        """y = CFB(cat(a, b)) written into the view `out` (20 channels)."""
        t, dev = self.torch, self.device
        tiles = n_chunks * ft_tiles(frames)
        sc = scratch if scratch is not None else {}
        def buf(key, ch, bins):
            if key not in sc or sc[key].tiles != tiles:
                sc[key] = FT(t, dev, n_chunks, frames, ch, bins)
            return sc[key]
        gx, r, li, hf, lo, ceps = (buf("gx", CH, F_BINS), buf("r", CH, F_BINS), buf("li", 2 * CH, CEPS_F),
                                   buf("hf", 2 * CH, CEPS_F), buf("lo", 2 * CH, CEPS_F), buf("ceps", CH, F_BINS))
        s0 = self.stats(a, b, F_BINS, tiles)
        self.pw(1, a, b, self._ln(s0, name + ".LN0"), name + ".conv_gate.weight", name + ".conv_gate.bias", gx.view(), F_BINS, CH,
                w2=name + ".conv_input.weight", bias2=name + ".conv_input.bias", out1=r.view(), tiles=tiles)
        s2 = self.stats(r.view(), None, F_BINS, tiles)
        _lib.check(self.lib.vadx_dfsmn_dft_f(0, C.byref(r.view()), None, C.byref(self._ln(s2, name + ".LN2")),
                                             self.tbl_fwd.data_ptr(), C.byref(li.view()), CH, tiles, _lib.stream_ptr()))
        sl = self.stats(li.view(), None, CEPS_F, tiles)
        self.lstm_f(name + ".ceps_unit.ch_lstm_f", li.view(), self._ln(sl, name + ".ceps_unit.LN"), hf.view(), CEPS_F, tiles)
        self.pw(0, hf.view(), None, None, name + ".ceps_unit.ch_lstm_f.linear.weight", name + ".ceps_unit.ch_lstm_f.linear.bias",
                lo.view(), CEPS_F, 2 * CH, tiles=tiles)
        _lib.check(self.lib.vadx_dfsmn_dft_f(1, C.byref(li.view()), C.byref(lo.view()), None, self.tbl_inv.data_ptr(),
                                             C.byref(ceps.view()), CH, tiles, _lib.stream_ptr()))
        s1 = self.stats(gx.view(), None, F_BINS, tiles)
        self.pw(2, gx.view(), None, self._ln(s1, name + ".LN1"), name + ".conv.weight", name + ".conv.bias", out, F_BINS, CH, kf=3,
                add=ceps.view(), tiles=tiles)
        return sc
