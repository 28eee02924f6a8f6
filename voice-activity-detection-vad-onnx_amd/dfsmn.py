"""DFSMN near+far VAD on MI355X (reference: DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py,
Inference_DFSMN_VAD_ONNX.py).  This module is plumbing: it owns the device weights/tables and strings
the libvadx kernels together in the order of `DFSMN_VAD.forward` / `NET.forward`; all arithmetic is HIP.

Activations use the frame-tiled "FT" layout documented in csrc/dfsmn.hip."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

F_BINS, CEPS_F, CH = 160, 81, 20


def ft_tiles(frames):
    return (frames + 15) // 16


STAT_PARTS = 2             # VADX_DFSMN_STAT_PARTS (include/vadx.h)


def packed_stride(frames):
    """Frame stride of the PACKED FT layout: the frames rounded up to 8, so that a chunk starts on a tile or a half tile (101 ->
    104: 6.5 tiles per window instead of 7) and the time-axis LSTMs' aligned 4-frame accesses never straddle a tile."""
    return (frames + 7) // 8 * 8


class FT:
    """A device tensor in FT layout + helpers to make channel-slice views."""

    def __init__(self, torch, device, n_chunks, frames, channels, bins, zero=True, stride=None):
        """zero=False for intermediates that a kernel overwrites completely (all 16 columns of every tile): the MFMA
        columns (frames) are independent, so whatever the padding frames hold never reaches a valid frame.
        stride: frames between the starts of two chunks (None: every chunk on tiles of its own, ft_tiles(frames) * 16)."""
        self.nt = ft_tiles(frames)
        self.stride = self.nt * 16 if stride is None else int(stride)
        self.tiles = (n_chunks * self.stride + 15) // 16
        self.C, self.F, self.frames = channels, bins, frames
        alloc = torch.zeros if zero else torch.empty
        self.data = alloc((self.tiles, channels, bins, 16), dtype=torch.float32, device=device)

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
    lib = property(lambda self: _lib.lib())

    def __init__(self, weights, device="cuda:0"):
        self.torch = t = _lib.require_gpu()
        self.device = t.device(device)
        _lib.lib()                 # load / symbol check now; `self.lib` resolves per call (so _lib.trace() sees the launches)
        w = {k[len("iccrn."):]: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in weights.items()
             if k.startswith("iccrn.")}
        self.w = w
        self.d = {}
        self._cfb_cache = {}
        self.arithmetic = None         # None = the module default (_lib.gemm_mode()); "f32" | "split" | "h2" (-> "split" in lstm_f / cfb, see _arith)
        self.range_flag = t.zeros(2, dtype=t.int32, device=self.device)      # fp16 x 2 kernels (the two-layer time LSTM): {sticky flag, max |x| bits}
        self.lstm_t_h2 = True          # the two LSTMs' fp16 x 2 forms; cleared by DfsmnEngine.run while it recomputes a flagged batch
        env = os.environ.get("VADX_CFB_BACK", "")     # read once: the second half of the gated conv block as split products (opt-in, no faster)
        self.cfb_back_split = env == "split"
        self.frame_stride = None       # None: every chunk on tiles of its own; forward() packs (packed_stride) for the duration of a pass

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
        # forward: 160 table rows = cos bins 0..80 | sin bins 1..79 (the sine rows of bins 0 and 80 are identically zero: the kernel
        # writes those two outputs as zeros), ten MFMA row tiles instead of 96 + 96 padded rows
        fwd = t.cat([cos_k, sin_k[1:half]], dim=0).contiguous()                       # [160, 160]
        fb = t.fft.fft(t.eye(n, dtype=t.float32))
        basis = t.vstack([t.real(fb[:half + 1]), t.imag(fb[:half + 1])]).float()
        inv_basis = t.linalg.pinv(basis).T                                            # [162 (re 0..80 | im 0..80), 160]
        # inverse: k = re 0..80 | im 1..79 -- the imaginary parts of bins 0 and 80 meet all-zero rows of the pseudo-inverse
        dead = inv_basis[[half + 1, 2 * half + 1]]
        if float(dead.abs().max()) > 1e-6:
            raise ValueError("CepsUnit inverse basis: the rows of im(bin 0) / im(bin 80) are not zero")
        inv = t.cat([inv_basis[:half + 1], inv_basis[half + 2:2 * half + 1]], dim=0).t().contiguous()      # [160 (f), 160 (k)]
        self.tbl_fwd, self.tbl_inv = fwd.to(self.device), inv.to(self.device)

    # ---- small helpers around the C ABI -------------------------------------------------------
    def _tiles(self, n_chunks, frames):
        return (n_chunks * (ft_tiles(frames) * 16 if self.frame_stride is None else self.frame_stride) + 15) // 16

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

    def new_part(self, tiles):
        """Buffer for the partial LayerNorm statistics a producing kernel emits (include/vadx.h, stats_merge)."""
        t = self.torch
        return t.empty((tiles, STAT_PARTS, 16, 4), dtype=t.float32, device=self.device)

    def merged_stats(self, part_a, part_b, tiles):
        """(mean, 1/(std + eps)) per frame of a tensor (or of cat(a, b)) from the partials its producers emitted."""
        t = self.torch
        s = t.empty((tiles, 16, 2), dtype=t.float32, device=self.device)
        _lib.check(self.lib.vadx_dfsmn_stats_merge(part_a.data_ptr(), None if part_b is None else part_b.data_ptr(), tiles,
                                                   s.data_ptr(), _lib.stream_ptr()))
        return s

    def pw(self, mode, a, b, ln, w, bias, out0, F, co, kf=1, act=0, w2=None, bias2=None, add=None, out1=None, tiles=None,
           part0=None, part1=None):
        _lib.check(self.lib.vadx_dfsmn_pw_conv(mode, C.byref(a), None if b is None else C.byref(b),
                                               None if ln is None else C.byref(ln), self._p(w), self._p(bias),
                                               None if w2 is None else self._p(w2), None if bias2 is None else self._p(bias2),
                                               None if add is None else C.byref(add), C.byref(out0),
                                               None if out1 is None else C.byref(out1), F, co, kf, act, tiles,
                                               None if part0 is None else part0.data_ptr(),
                                               None if part1 is None else part1.data_ptr(), _lib.stream_ptr()))

    def _arith(self):
        """VADX_ARITH_* of the kernels that have a split form (lstm_f, cfb_front): the engine's / module's mode, "h2" mapped to bf16 x 3 --
        the fp16 x 2 forms of these two kernels are not built yet."""
        m = self.arithmetic or _lib.gemm_mode()
        return _lib.ARITH["f32"] if m == "f32" else _lib.ARITH["split"]

    def lstm_f(self, prefix, inp, ln, out, F, tiles):
        arr = lambda n: (C.c_void_p * 2)(self._p(f"{prefix}.lstm2.{n}_l0"), self._p(f"{prefix}.lstm2.{n}_l0_reverse"))   # noqa: E731
        wi, wh, bi, bh = arr("weight_ih"), arr("weight_hh"), arr("bias_ih"), arr("bias_hh")
        h2 = (self.arithmetic or _lib.gemm_mode()) == "h2" and self.lstm_t_h2 and inp.c == 40       # the CepsUnit form has the fp16 x 2 kernel
        _lib.check(self.lib.vadx_dfsmn_lstm_f(C.byref(inp), None if ln is None else C.byref(ln), C.byref(wi), C.byref(wh),
                                              C.byref(bi), C.byref(bh), C.byref(out), F, tiles, _lib.stream_ptr(),
                                              _lib.ARITH["h2"] if h2 else self._arith(), self.range_flag.data_ptr()))

    # ---- CFB (:76-93) ---------------------------------------------------------------------------
    def _cfb_tables(self, name):
        """Device-side `vadx_dfsmn_cfb_weights` of one block (include/vadx.h): the block's own weights plus the tables the two
        streaming kernels derive from them -- DFT tables in MFMA fragment order, CepsUnit's Linear with (re, im) rows paired per
        lane, and the LayerNorm corrections TW = T w2, TB = T b2, CW = conv31(w1), CB = conv31(b1) + bias (float64 on the host)."""
        if name in self._cfb_cache:
            return self._cfb_cache[name]
        t, w = self.torch, self.w
        keep = []

        def dev(a):
            d = t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
            keep.append(d)
            return d.data_ptr()

        def frag(tbl, ksteps, col):
            """[160][*] table -> [10 row tiles][ksteps][64 lanes]: lane (q, i) of (m, s) = tbl[16 m + i][col(s, q)] (col < 0: zero)"""
            out = np.zeros((10, ksteps, 64), np.float64)
            for s_ in range(ksteps):
                for q in range(4):
                    k = col(s_, q)
                    if k >= 0:
                        out[:, s_, 16 * q:16 * q + 16] = tbl[:, k].reshape(10, 16)
            return out

        def fix(a, b):
            """two [20][160] tables -> [20][10][64]: lane quarter 0 = a, quarter 1 = b, rest zero"""
            out = np.zeros((CH, 10, 64), np.float64)
            out[:, :, 0:16] = a.reshape(CH, 10, 16)
            out[:, :, 16:32] = b.reshape(CH, 10, 16)
            return out

        def qfrag(tbl):
            """[160][K] float32 table -> bf16 x 3 split A fragments [10 row tiles][K / 32 chunks][3 planes][256 floats] (csrc/split3.h):
            x = t0 + t1 + t2 exactly, by truncation; lane 16 g + i of a fragment holds tbl[16 tile + i][32 chunk + 8 g + e], e = 0..7"""
            x = np.ascontiguousarray(tbl, dtype=np.float32)
            chunks = x.shape[1] // 32
            planes = []
            for _ in range(3):
                hi = (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
                planes.append((hi.view(np.uint32) >> 16).astype(np.uint16))
                x = x - hi                                                    # exact in float32
            pl = np.stack(planes, 0).reshape(3, 10, 16, chunks, 4, 8)        # [plane][tile][i][chunk][g][e]
            out = np.ascontiguousarray(np.transpose(pl, (1, 3, 0, 4, 2, 5)))  # [tile][chunk][plane][g][i][e]
            return out.reshape(10, chunks, 3, 512).view(np.float32)

        fwd = self.tbl_fwd.cpu().numpy().astype(np.float64)                  # [160 rows][160 f]
        inv = self.tbl_inv.cpu().numpy().astype(np.float64)                  # [160 f][160 k: re 0..80 | im 1..79]
        def inv_col(s_, q):
            if s_ < 21:
                k = 4 * s_ + q
                return k if k <= 80 else -1
            k = 4 * (s_ - 21) + q                                             # imaginary part of bin k
            return 80 + k if 1 <= k <= 79 else -1
        ln1_w, ln1_b = w[name + ".LN1.w"].reshape(CH, F_BINS).astype(np.float64), w[name + ".LN1.b"].reshape(CH, F_BINS).astype(np.float64)
        ln2_w, ln2_b = w[name + ".LN2.w"].reshape(CH, F_BINS).astype(np.float64), w[name + ".LN2.b"].reshape(CH, F_BINS).astype(np.float64)
        w31 = w[name + ".conv.weight"][:, :, :, 0].astype(np.float64)       # [co][ci][tap]
        def conv31(x):                                                        # [ci][160] -> [co][160], zero padding
            xp = np.pad(x, ((0, 0), (1, 1)))
            return sum(w31[:, :, tap] @ xp[:, tap:tap + F_BINS] for tap in range(3))
        lin_w, lin_b = w[name + ".ceps_unit.ch_lstm_f.linear.weight"], w[name + ".ceps_unit.ch_lstm_f.linear.bias"]
        lw, lb = np.zeros((48, 2 * CH), np.float32), np.zeros(48, np.float32)
        for row in range(48):
            tile, qq, rr = row // 16, (row % 16) // 4, row % 4
            c = 8 * tile + 2 * qq + (rr & 1)
            if c < CH:
                lw[row], lb[row] = lin_w[(rr >> 1) * CH + c], lin_b[(rr >> 1) * CH + c]
        cw = _lib.DfsmnCfbWeights()
        cw.ln0_w = self._p(name + ".LN0.w")
        cw.gate_w, cw.in_w, cw.in_b = self._p(name + ".conv_gate.weight"), self._p(name + ".conv_input.weight"), self._p(name + ".conv_input.bias")
        cw.conv_w = self._p(name + ".conv.weight")
        # LayerNorm 0 behind the gate conv: Wg (w0 * x) scaled by inv0, plus Wg b0 + bias, minus mean0 inv0 (Wg w0)
        wg = w[name + ".conv_gate.weight"][:, :, 0, 0].astype(np.float64)                 # [20][cin]
        cin = wg.shape[1]
        ln0_w, ln0_b = w[name + ".LN0.w"].reshape(cin, F_BINS).astype(np.float64), w[name + ".LN0.b"].reshape(cin, F_BINS).astype(np.float64)
        tab = np.stack([wg @ ln0_w, wg @ ln0_b + w[name + ".conv_gate.bias"].astype(np.float64)[:CH, None], ln1_w, ln2_w], 0)      # [4][20][160]
        cw.front_tab = dev(np.transpose(tab, (2, 0, 1)))                                  # [160][4][20]
        cw.fwd_tbl = dev(frag(fwd, 40, lambda s_, q: 4 * s_ + q))
        cw.fwd_fix = dev(fix(ln2_w @ fwd.T, ln2_b @ fwd.T))
        cw.fwd_tbl_q = dev(qfrag(self.tbl_fwd.cpu().numpy()))
        cw.lin_w, cw.lin_b = dev(lw), dev(lb)
        cw.inv_tbl = dev(frag(inv, 41, inv_col))
        # split-product kernel: k-step j = [re of ceps bins 16 j .. 16 j + 15 | im of the same bins]; im of bin 0 does not exist, its
        # slot carries re of bin 80 (160 = 81 + 79 values: five k-steps without padding)
        perm = np.empty(F_BINS, np.int64)
        for j in range(5):
            for s_ in range(32):
                b_ = 16 * j + (s_ & 15)
                perm[32 * j + s_] = b_ if s_ < 16 else (80 if b_ == 0 else 80 + b_)
        assert sorted(perm.tolist()) == list(range(F_BINS))
        cw.inv_tbl_q = dev(qfrag(self.tbl_inv.cpu().numpy()[:, perm]))
        cw.out_fix = dev(fix(conv31(ln1_w), conv31(ln1_b) + w[name + ".conv.bias"].astype(np.float64)[:CH, None]))
        self._cfb_cache[name] = (cw, keep)
        return self._cfb_cache[name]

    def cfb(self, name, a, b, out, n_chunks, frames, scratch=None, a_part=None, b_part=None):
        """y = CFB(cat(a, b)) written into the view `out` (20 channels): cfb_front -> lstm_f -> cfb_back (csrc/dfsmn_cfb.hip).
        a_part / b_part: the partial statistics the producers of a / b emitted (None: a frame_stats pass over the tensor);
        returns (scratch, partial statistics of y)."""
        t, dev = self.torch, self.device
        tiles = self._tiles(n_chunks, frames)
        sc = scratch if scratch is not None else {}
        def buf(key, ch, bins):
            if key not in sc or sc[key].tiles != tiles:
                sc[key] = FT(t, dev, n_chunks, frames, ch, bins, zero=False, stride=self.frame_stride)
            return sc[key]
        y1, li, hf = buf("y1", CH, F_BINS), buf("li", 2 * CH, CEPS_F), buf("hf", 2 * CH, CEPS_F)
        def sbuf(key):
            if key not in sc or sc[key].shape[0] != tiles:
                sc[key] = t.empty((tiles, 16, 2), dtype=t.float32, device=dev)
            return sc[key]
        s1, sl = sbuf("stats1"), sbuf("stats_li")
        out_part = self.new_part(tiles)                 # outlives the block (encoder outputs are normalised again by the decoder)
        have = a_part is not None and (b is None or b_part is not None)
        s0 = self.merged_stats(a_part, b_part if b is not None else None, tiles) if have else self.stats(a, b, F_BINS, tiles)
        cw, _keep = self._cfb_tables(name)
        cw.front_arithmetic = self._arith()
        cw.back_arithmetic = _lib.ARITH["split"] if (self.cfb_back_split and self._arith() == _lib.ARITH["split"]) else _lib.ARITH["f32"]
        st = _lib.stream_ptr()
        _lib.check(self.lib.vadx_dfsmn_cfb_front(C.byref(cw), C.byref(a), None if b is None else C.byref(b), s0.data_ptr(),
                                                 y1.data.data_ptr(), s1.data_ptr(), li.data.data_ptr(), sl.data_ptr(), tiles, st))
        self.lstm_f(name + ".ceps_unit.ch_lstm_f", li.view(), self._ln(sl, name + ".ceps_unit.LN"), hf.view(), CEPS_F, tiles)
        _lib.check(self.lib.vadx_dfsmn_cfb_back(C.byref(cw), hf.data.data_ptr(), li.data.data_ptr(), y1.data.data_ptr(), s1.data_ptr(),
                                                C.byref(out), out_part.data_ptr(), tiles, st))
        return sc, out_part

    def cfb_unfused(self, name, a, b, out, n_chunks, frames, scratch=None, a_part=None, b_part=None):
        """The same block as six launches of the building-block kernels (pw_conv / dft_f / lstm_f, csrc/dfsmn.hip) with every
        intermediate in HBM: the A/B reference for `cfb` (tests/test_gpu_dfsmn.py) -- not used by the engine."""
        t, dev = self.torch, self.device
        tiles = n_chunks * ft_tiles(frames)
        sc = scratch if scratch is not None else {}
        def buf(key, ch, bins):
            if key not in sc or sc[key].tiles != tiles:
                sc[key] = FT(t, dev, n_chunks, frames, ch, bins, zero=False)
            return sc[key]
        gx, r, li, hf, lo, ceps = (buf("gx", CH, F_BINS), buf("r", CH, F_BINS), buf("li", 2 * CH, CEPS_F),
                                   buf("hf", 2 * CH, CEPS_F), buf("lo", 2 * CH, CEPS_F), buf("ceps", CH, F_BINS))
        def pbuf(key):
            if key not in sc or sc[key].shape[0] != tiles:
                sc[key] = self.new_part(tiles)
            return sc[key]
        gx_part, r_part, li_part = pbuf("gx_part"), pbuf("r_part"), pbuf("li_part")
        out_part = self.new_part(tiles)                 # outlives the block (encoder outputs are normalised again by the decoder)
        have = a_part is not None and (b is None or b_part is not None)
        s0 = self.merged_stats(a_part, b_part if b is not None else None, tiles) if have else self.stats(a, b, F_BINS, tiles)
        self.pw(1, a, b, self._ln(s0, name + ".LN0"), name + ".conv_gate.weight", name + ".conv_gate.bias", gx.view(), F_BINS, CH,
                w2=name + ".conv_input.weight", bias2=name + ".conv_input.bias", out1=r.view(), tiles=tiles,
                part0=gx_part, part1=r_part)
        s2 = self.merged_stats(r_part, None, tiles)
        _lib.check(self.lib.vadx_dfsmn_dft_f(0, C.byref(r.view()), None, C.byref(self._ln(s2, name + ".LN2")),
                                             self.tbl_fwd.data_ptr(), C.byref(li.view()), CH, tiles, li_part.data_ptr(),
                                             _lib.stream_ptr()))
        sl = self.merged_stats(li_part, None, tiles)
        self.lstm_f(name + ".ceps_unit.ch_lstm_f", li.view(), self._ln(sl, name + ".ceps_unit.LN"), hf.view(), CEPS_F, tiles)
        self.pw(0, hf.view(), None, None, name + ".ceps_unit.ch_lstm_f.linear.weight", name + ".ceps_unit.ch_lstm_f.linear.bias",
                lo.view(), CEPS_F, 2 * CH, tiles=tiles)
        _lib.check(self.lib.vadx_dfsmn_dft_f(1, C.byref(li.view()), C.byref(lo.view()), None, self.tbl_inv.data_ptr(),
                                             C.byref(ceps.view()), CH, tiles, None, _lib.stream_ptr()))
        s1 = self.merged_stats(gx_part, None, tiles)
        self.pw(2, gx.view(), None, self._ln(s1, name + ".LN1"), name + ".conv.weight", name + ".conv.bias", out, F_BINS, CH, kf=3,
                add=ceps.view(), tiles=tiles, part0=out_part)
        return sc, out_part

    # ---- LSTMs along time (:252-267) ------------------------------------------------------------
    def lstm_t(self, which, prefix, inp, ln, mul, out, frames, n_chunks):
        layers = 2 if which == 0 else 1
        arr = lambda n: (C.c_void_p * 2)(*[self._p(f"{prefix}.lstm2.{n}_l{l}") if l < layers else None for l in range(2)])   # noqa: E731
        wi, wh, bi, bh = arr("weight_ih"), arr("weight_hh"), arr("bias_ih"), arr("bias_hh")
        stride = ft_tiles(frames) * 16 if self.frame_stride is None else self.frame_stride
        h2 = (self.arithmetic or _lib.gemm_mode()) == "h2" and self.lstm_t_h2          # (only the two-layer net has the fp16 x 2 form)
        _lib.check(self.lib.vadx_dfsmn_lstm_t_ex(which, C.byref(inp), None if ln is None else C.byref(ln), C.byref(wi), C.byref(wh),
                                                 C.byref(bi), C.byref(bh), self._p(prefix + ".linear.weight"),
                                                 self._p(prefix + ".linear.bias"), None if mul is None else C.byref(mul),
                                                 C.byref(out), F_BINS, frames, n_chunks, stride, _lib.stream_ptr(),
                                                 _lib.ARITH["h2" if h2 else "f32"], self.range_flag.data_ptr()))

    # ---- NET.forward (:226-249) without the ISTFT ------------------------------------------------
    def forward(self, x4, n_chunks, frames, pack=True):
        """x4: FT (4 ch: mix re, mix im, scaled far re, scaled far im) -> Y FT (2 ch: re, im of the AEC spectrum).
        pack: run the network on the PACKED frame layout (chunks 104 frames apart instead of on 7 tiles = 112 columns each: the
        per-frame kernels see 7 % fewer tiles); only the 4-channel input and the 2-channel output are repacked."""
        if pack and packed_stride(frames) < ft_tiles(frames) * 16:
            t, st = self.torch, _lib.stream_ptr()
            stride = packed_stride(frames)
            xp = FT(t, self.device, n_chunks, frames, 4, F_BINS, zero=False, stride=stride)
            _lib.check(self.lib.vadx_dfsmn_ft_repack(x4.data.data_ptr(), xp.data.data_ptr(), 4, F_BINS, frames, n_chunks, x4.stride, stride, st))
            self.frame_stride = stride
            try:
                yp = self.forward(xp, n_chunks, frames, pack=False)
            finally:
                self.frame_stride = None
            y = FT(t, self.device, n_chunks, frames, 2, F_BINS, zero=False)
            _lib.check(self.lib.vadx_dfsmn_ft_repack(yp.data.data_ptr(), y.data.data_ptr(), 2, F_BINS, frames, n_chunks, stride, y.stride, st))
            return y
        t, dev = self.torch, self.device
        tiles = self._tiles(n_chunks, frames)
        new = lambda ch, bins=F_BINS: FT(t, dev, n_chunks, frames, ch, bins, zero=False, stride=self.frame_stride)      # noqa: E731
        hf0, e0l = new(2 * CH), new(CH)
        cats = [new(2 * CH) for _ in range(5)]          # [e0|d1], [e1|d2], [e2|d3], [e3|d4], [e4|d5]
        e5, p5, d0, y = new(CH), new(CH), new(2 * CH), new(2)
        self.lstm_f("in_ch_lstm", x4.view(), None, hf0.view(), F_BINS, tiles)
        self.pw(0, hf0.view(), None, None, "in_ch_lstm.linear.weight", "in_ch_lstm.linear.bias", e0l.view(), F_BINS, CH, tiles=tiles)
        e_part = [self.new_part(tiles)]                 # partial statistics of e0..e4 (each is normalised twice: by the next
        self.pw(0, e0l.view(), x4.view(), None, "in_conv.weight", "in_conv.bias", cats[0].view(0, CH), F_BINS, CH, tiles=tiles,
                part0=e_part[0])                        # encoder block and, concatenated with d_{k+1}, by decoder block k)
        sc = {}
        for k in range(1, 5):                           # e1..e4
            sc, pk = self.cfb(f"cfb_e{k}", cats[k - 1].view(0, CH), None, cats[k].view(0, CH), n_chunks, frames, sc,
                              a_part=e_part[k - 1])
            e_part.append(pk)
        sc, e5_part = self.cfb("cfb_e5", cats[4].view(0, CH), None, e5.view(), n_chunks, frames, sc, a_part=e_part[4])
        s5 = self.merged_stats(e5_part, None, tiles)
        self.lstm_t(0, "ch_lstm", e5.view(), self._ln(s5, "ln"), e5.view(), p5.view(), frames, n_chunks)
        sc, d_part = self.cfb("cfb_d5", p5.view(), None, cats[4].view(CH, CH), n_chunks, frames, sc)     # p5: frame_stats pass
        for k in range(4, 0, -1):                       # d4..d1: cfb_dk(cat[e_k, d_{k+1}]) -> second half of cats[k-1]
            sc, d_part = self.cfb(f"cfb_d{k}", cats[k].view(0, CH), cats[k].view(CH, CH), cats[k - 1].view(CH, CH), n_chunks, frames,
                                  sc, a_part=e_part[k], b_part=d_part)
        self.lstm_t(1, "out_ch_lstm", cats[0].view(), None, None, d0.view(), frames, n_chunks)
        self.pw(0, d0.view(), cats[0].view(CH, CH), None, "out_conv.weight", "out_conv.bias", y.view(), F_BINS, 2, tiles=tiles)
        return y


class DfsmnEngine:
    """Batched DFSMN near+far VAD: two int16 streams in, vad_results [51] per 16001-sample window out."""
    lib = property(lambda self: _lib.lib())

    L, T_B, T_A, LOOK_BACKWARD, FRAME = 16001, 101, 51, 0.3, 320

    def __init__(self, weights=None, device="cuda:0", sub_batch=256):
        from . import tables, weights as _w, frontend as _fe
        self.torch = t = _lib.require_gpu()
        self.device = t.device(device)
        _lib.lib()                 # load / symbol check now; `self.lib` resolves per call (so _lib.trace() sees the launches)
        from . import checkpoints as _ck
        w = _ck.resolve("dfsmn", weights)
        w = {k: np.ascontiguousarray(np.asarray(v), dtype=np.float32) for k, v in w.items()}
        self.sub_batch = int(sub_batch)
        self.range_fallbacks = 0       # batches recomputed because the fp16 x 2 time LSTM flagged an out-of-range operand
        self.iccrn = Iccrn(w, device)
        dev = lambda a: t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)     # noqa: E731
        self.alpha = [dev(w[k]) for k in ("alpha.linear1.weight", "alpha.linear1.bias", "alpha.linear2.weight", "alpha.linear2.bias")]
        # ISTFT buffers of NET.__init__ (:183-207), float32 torch ops in the reference's order
        n, hop, half = 319, 160, 159
        window = t.hamming_window(n)
        fe_ = t.fft.fft(t.eye(n, dtype=t.float32))
        fb = t.vstack([t.real(fe_[:half + 1]), t.imag(fe_[:half + 1])]).float()
        inverse_basis = t.linalg.pinv((fb * n) / hop).T * window.view(1, -1)          # [320 ch][319]
        bt = t.zeros((320, 320), dtype=t.float32)
        bt[:319, :] = inverse_basis.t()
        max_frames = 200
        out_len = (max_frames - 1) * hop + n
        ws = t.zeros(out_len, dtype=t.float32)
        wsq = window ** 2
        for i in range(max_frames):
            s0 = i * hop
            ln_ = min(n, out_len - s0)
            if ln_ <= 0:
                break
            ws[s0:s0 + ln_] += wsq[:ln_]
        self.basis_t, self.wsum_inv = bt.to(self.device), (n / (ws * hop + 1e-6)).to(self.device)
        # front-ends: STFT-B (hamming 319/319/160, centre pad 159) raw complex; STFT-A (hamming 640 in 1024, hop 320) + htk fbank
        inv = 1.0 / 32768.0
        self.fe_b = _fe.Frontend(dict(n_fft=319, win=319, hop=160, window="hamming", variant="v1b", center=True, prep=2,
                                      k=(0.0, inv), mel=("zeros",), log_mode=0, log_floor=1e-6), self.L, device=device)
        mk = lambda prep: _fe.Frontend(dict(n_fft=1024, win=640, hop=320, window="hamming", variant="v1b", center=True, prep=prep,   # noqa: E731
                                            k=(1.15, inv), mel=("torchaudio", 20, 8000, None, "htk"), log_mode=0, log_floor=1e-6),
                                       self.L, device=device)
        self.fe_a = {3: mk(3), 4: mk(4), 5: mk(5)}
        m = dict(_w.DFSMN_MASK)
        m["hidden"] = w["mask.linear1.weight"].shape[0]
        m["layers"] = len([k for k in w if k.startswith("mask.deepfsmn.") and k.endswith(".linear.weight")])
        if m["layers"]:
            m["fsmn_hidden"] = w["mask.deepfsmn.0.linear.weight"].shape[0]
            m["lorder"] = w["mask.deepfsmn.0.conv1.weight"].shape[2]
        mw = _lib.DfsmnMaskWeights()
        mw.hidden, mw.fsmn_hidden, mw.layers, mw.lorder = m["hidden"], m["fsmn_hidden"], m["layers"], m["lorder"]
        self._mask_keep = []

        def mdev(a, pad_axes=(), frag=False):
            a = np.asarray(a, dtype=np.float32)
            pads = [(0, (-a.shape[ax]) % 16 if ax in pad_axes else 0) for ax in range(a.ndim)]
            d_ = dev(_lib.frag_major(a) if frag else np.pad(a, pads))       # GEMM operands: fragment-major
            self._mask_keep.append(d_)
            return d_.data_ptr()
        shift = w["mask.shift"] + np.float32(np.log(np.float32(32768.0 ** 2)))      # wrapper __init__ :291
        mw.shift, mw.scale = mdev(shift), mdev(w["mask.scale"])
        mw.linear1_w, mw.linear1_b = mdev(w["mask.linear1.weight"], frag=True), mdev(w["mask.linear1.bias"], (0,))
        mw.linear3_w, mw.linear3_b = mdev(w["mask.linear3.weight"].reshape(-1)), mdev(w["mask.linear3.bias"])
        for l in range(m["layers"]):
            mw.fsmn_linear_w[l] = mdev(w[f"mask.deepfsmn.{l}.linear.weight"], frag=True)
            mw.fsmn_linear_b[l] = mdev(w[f"mask.deepfsmn.{l}.linear.bias"], (0,))
            mw.fsmn_project_w[l] = mdev(w[f"mask.deepfsmn.{l}.project.weight"], frag=True)
            mw.fsmn_conv_w[l] = mdev(w[f"mask.deepfsmn.{l}.conv1.weight"][:, 0, :, 0])
        self.mask = mw

    def set_near_only_constants(self, pow_far, far_comp):
        """The near-end-only model (DFSMN/only_near_end_audio/Export_DFSMN_VAD.py:309-310, :327-331) replaces the far-end
        spectrum by two tensors of white noise drawn ONCE at export and baked into the graph: pow_far [160, T>=T_B, 10]
        and far_comp [2, 160, T>=T_B] (float16 values).  They are part of that model's weights: pass the ones of the
        export being replaced, or `weights.dfsmn_near_only_constants(seed)` for a stand-in."""
        t = self.torch
        pf = np.asarray(pow_far, dtype=np.float32)[:, :self.T_B, :]
        fc = np.asarray(far_comp, dtype=np.float32)[:, :, :self.T_B]
        if pf.shape != (F_BINS, self.T_B, 10) or fc.shape != (2, F_BINS, self.T_B):
            raise ValueError(f"near-only constants must cover [160, {self.T_B}, 10] and [2, 160, {self.T_B}]")
        self.pow_far = t.from_numpy(np.ascontiguousarray(pf)).to(self.device)
        self.far_ft = to_ft(t, t.from_numpy(np.ascontiguousarray(fc)).unsqueeze(0), self.device).data      # [nt, 2, 160, 16]

    def run(self, near_i16, far_i16=None, windows_per_clip=1, win_stride=None, return_aec=False):
        """near/far int16 [B, N] on the window grid -> vad f32 [B*W, 51] (each window stateless, as the reference).
        far_i16 None = the near-end-only model (needs set_near_only_constants)."""
        t = self.torch
        to_dev = lambda a: (a if t.is_tensor(a) else t.from_numpy(np.ascontiguousarray(a, dtype=np.int16))).to(self.device).contiguous()   # noqa: E731
        if far_i16 is None and getattr(self, "pow_far", None) is None:
            raise ValueError("near-only run: call set_near_only_constants(pow_far, far_comp) first")
        near = to_dev(near_i16)
        far = None if far_i16 is None else to_dev(far_i16)
        B, W = near.shape[0], int(windows_per_clip)
        ws = self.L if win_stride is None else int(win_stride)
        n_chunks = B * W
        vad = t.empty((n_chunks, self.T_A), dtype=t.float32, device=self.device)
        aec_all = t.empty((n_chunks, self.L), dtype=t.float32, device=self.device) if return_aec else None
        per = max(1, self.sub_batch // W)                  # clips per sub-batch (activations are ~30 MB per window)
        # (alternating the independent sub-batches over two or three side streams so that one's HBM-bound launches overlap another's
        # matrix-pipe-bound ones was measured: 2986 / 2985 ms against 2990 -- every launch fills the chip by itself)
        def sweep():
            for b0 in range(0, B, per):
                nb = min(per, B - b0)
                v, a = self._run_sub(near[b0:b0 + nb], None if far is None else far[b0:b0 + nb], W, ws)
                vad[b0 * W:(b0 + nb) * W] = v
                if return_aec:
                    aec_all[b0 * W:(b0 + nb) * W] = a
        sweep()
        net = self.iccrn
        if (net.arithmetic or _lib.gemm_mode()) == "h2" and int(net.range_flag[0].item()) != 0:
            # an input of the fp16 x 2 time LSTM left the fp16 range (one 4-byte read-back, synchronises): the batch again on float32 MFMAs
            self.range_fallbacks += 1
            net.range_flag.zero_()
            net.lstm_t_h2 = False
            try:
                sweep()
            finally:
                net.lstm_t_h2 = True
        return (vad, aec_all) if return_aec else vad

    def run_from_host(self, host_near_i16, host_far_i16, windows_per_clip=1, win_stride=None, chunk_clips=64, feed=None):
        """`run` fed from HOST memory (two int16 [B, N] tensors, ideally pinned: vadx.feed.pin), uploads overlapped with compute
        (vadx.feed.HostPcmFeed with two streams); bit-identical to `run` of the resident batch when chunk_clips * windows_per_clip
        is a multiple of the engine's sub-batch cut or below it (the same launches see the same windows)."""
        from . import feed as _feed
        f = feed or _feed.HostPcmFeed(self.device, host_near_i16.shape[1], chunk_clips, streams=2)
        return _feed.cat_results(f.map([host_near_i16, host_far_i16], lambda a, b: self.run(a, b, windows_per_clip, win_stride)))

    def _run_sub(self, near, far, W, ws):
        t, lib = self.torch, self.lib
        B = near.shape[0]
        n = B * W
        nt = ft_tiles(self.T_B)
        xraw = FT(t, self.device, n, self.T_B, 4, F_BINS)
        x4 = FT(t, self.device, n, self.T_B, 4, F_BINS)
        mean_near = t.empty((n,), dtype=t.float32, device=self.device)
        mean_far = t.empty((n,), dtype=t.float32, device=self.device)
        st = _lib.stream_ptr()
        with t.cuda.device(self.device):
            for src, means, coff in ((near, mean_near, 0), (far, mean_far, 2)):
                if src is None:          # near-only: channels 2, 3 = the baked far spectrum, the same for every window
                    xraw.data.view(n, nt, 4, F_BINS, 16)[:, :, 2:4] = self.far_ft.unsqueeze(0)
                    continue
                _lib.check(lib.vadx_frontend_stft_ft(C.byref(self.fe_b.cfg), self.fe_b.packed.data_ptr(), src.data_ptr(),
                                                     _lib.row_stride(src), ws, B, W, means.data_ptr(), xraw.data.data_ptr(), 4, coff, st))
            _lib.check(lib.vadx_dfsmn_alpha_scale(xraw.data.data_ptr(), x4.data.data_ptr(), n, nt, *[a.data_ptr() for a in self.alpha],
                                                  None if far is not None else self.pow_far.data_ptr(), self.T_B, st))
            y = self.iccrn.forward(x4, n, self.T_B)
            z = t.empty((n, self.T_B, 320), dtype=t.float32, device=self.device)
            aec = t.empty((n, self.L), dtype=t.float32, device=self.device)
            _lib.check(lib.vadx_dfsmn_istft(y.data.data_ptr(), self.basis_t.data_ptr(), self.wsum_inv.data_ptr(), z.data_ptr(),
                                            aec.data_ptr(), n, self.T_B, st))
            feat = t.empty((n, self.T_A, 240), dtype=t.float32, device=self.device)
            for prep, off in ((3, 0), (4, 80), (5, 160)):
                fe = self.fe_a[prep]
                _lib.check(lib.vadx_frontend_logmel_ex(C.byref(fe.cfg), fe.packed.data_ptr(), fe.mel_kb.ctypes.data,
                                                       None if prep == 4 else near.data_ptr(), _lib.row_stride(near), ws, B, W,
                                                       None if prep == 4 else mean_near.data_ptr(),
                                                       None if prep == 3 else aec.data_ptr(), 240, off, feat.data_ptr(), st))
            vad = t.empty((n, self.T_A), dtype=t.float32, device=self.device)
            _lib.check(lib.vadx_dfsmn_mask_net(C.byref(self.mask), feat.data_ptr(), n, self.T_A, vad.data_ptr(), st))
        return vad, aec

    def grid(self):
        lb = int(self.LOOK_BACKWARD * 16000 // self.FRAME)
        return lb, self.L - (lb + 1) * self.FRAME

    def detect(self, near_clips, far_clips, pad_noise_near=None, pad_noise_far=None, fusion_threshold=0.3,
               min_speech_duration=0.2, speaking_score=0.5, silence_score=0.5):
        """Equal-length clip pairs (host, any numeric dtype) [B,N] -> per clip [(start_s, end_s)]
        (Inference_DFSMN_VAD_ONNX.py:124-163 prep, :221-278 loop)."""
        from . import timestamps as ts
        from .fsmn import pad_to_window_grid
        t = self.torch
        near_clips = np.asarray(near_clips)
        far_clips = None if far_clips is None else np.asarray(far_clips)      # None: the near-end-only model
        B = near_clips.shape[0]
        n = near_clips.shape[1] if far_clips is None else min(near_clips.shape[1], far_clips.shape[1])
        lb, stride = self.grid()
        rows_n, rows_f = [], []
        for b in range(B):
            a = ts.normalize_to_int16(near_clips[b, :n].astype(np.float32))
            rows_n.append(pad_to_window_grid(a, self.L, stride, None if pad_noise_near is None else pad_noise_near[b]))
            if far_clips is not None:
                f = ts.normalize_to_int16(far_clips[b, :n].astype(np.float32))
                rows_f.append(pad_to_window_grid(f, self.L, stride, None if pad_noise_far is None else pad_noise_far[b]))
        near, far = np.stack(rows_n), (np.stack(rows_f) if far_clips is not None else None)
        W = (near.shape[1] - self.L) // stride + 1
        vad = self.run(near, far, W, stride)
        flags = t.empty((B, W * (self.T_A - lb) + lb), dtype=t.uint8, device=self.device)
        with t.cuda.device(self.device):
            _lib.check(self.lib.vadx_dfsmn_vote(vad.data_ptr(), B, W, self.T_A, lb, float(speaking_score), float(silence_score),
                                                flags.data_ptr(), _lib.stream_ptr()))
        fl = flags.cpu().numpy().astype(bool)
        return [ts.process_timestamps(ts.vad_to_timestamps(fl[b], self.FRAME / 16000), fusion_threshold, min_speech_duration)
                for b in range(B)]


class DfsmnSession:
    """{'near_end_audio','far_end_audio': int16 [1,1,16001]} -> [vad_results f32 [51]]
    (DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py:377-393).  near_only = (pow_far, far_comp): the near-end-only export
    instead -- {'audio': int16 [1,1,16001]} (DFSMN/only_near_end_audio/Export_DFSMN_VAD.py:373-392)."""

    def __init__(self, weights=None, device="cuda:0", near_only=None, io_dtype="float32"):
        """io_dtype="float16": I/O-compatible with the reference's fp16 conversion (Optimize_ONNX.py:40-44, keep_io_types=False: vad_results
        leaves as float16); the arithmetic here stays float32, the scores are rounded once on the way out."""
        from .fsmn import _Meta
        if io_dtype not in ("float32", "float16"):
            raise ValueError("io_dtype must be 'float32' or 'float16'")
        self.io_dtype = np.float16 if io_dtype == "float16" else np.float32
        self.engine = DfsmnEngine(weights, device)
        self.near_only = near_only is not None
        if self.near_only:
            self.engine.set_near_only_constants(*near_only)
            self._inputs_meta = [_Meta("audio", [1, 1, 16001], "tensor(int16)")]
        else:
            self._inputs_meta = [_Meta("near_end_audio", [1, 1, 16001], "tensor(int16)"), _Meta("far_end_audio", [1, 1, 16001], "tensor(int16)")]
        self._outputs_meta = [_Meta("vad_results", [51], "tensor(float16)" if io_dtype == "float16" else "tensor(float)")]

    def get_inputs(self):
        return list(self._inputs_meta)

    def get_outputs(self):
        return list(self._outputs_meta)

    def get_providers(self):
        return ["VadxMI355XExecutionProvider"]

    def run(self, output_names, feeds):
        arrs = [np.asarray(feeds[m.name]) for m in self._inputs_meta]
        for a in arrs:
            if a.dtype != np.int16:
                raise ValueError("Unexpected input data type, expected: (tensor(int16))")
            if a.shape[-1] != 16001:
                raise ValueError("Got invalid dimensions for input: expected last dim 16001")
        near = arrs[0].reshape(-1, 16001)
        far = None if self.near_only else arrs[1].reshape(-1, 16001)
        vad = self.engine.run(near, far).cpu().numpy().astype(self.io_dtype)
        return [vad.reshape(-1) if vad.shape[0] == 1 else vad]
