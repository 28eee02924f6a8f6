#!/usr/bin/env python
"""The other BASELINE configs (C3 FSMN, C4 MarbleNet, C5 FireRed + DFSMN near+far), measured in the same process as
bench.py's headline and reported in its `secondary` object (bench.py imports this module; run it directly for the
secondary numbers alone:  python bench_models.py [--reps 3] [--only fsmn,marblenet,firered,dfsmn]).

Every workload: full BASELINE size, synthetic int16 burst clips generated on the GPU (resident in HBM before the timed
region), seeded synthetic weights of the reference architectures, device ms = median of `reps` passes bracketed by HIP
events on the launch stream; one extra traced pass (`vadx._lib.trace`) splits the time by C-ABI entry point; the
dominant entry gets a `roofline` (algorithmic flops of that stage as the reference computes them / its device time /
the f32-MFMA peak) and the whole pass gets `hbm` = SURVEY 8(d) algorithmic bytes / time / 8 TB/s.  `cpu_baseline` =
the torch-CPU oracle driven like the reference drives ORT (batch 1, one call per window), bounded to a few seconds.
"""
from __future__ import annotations

import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = the f32 vector rate
PEAK_HBM_GBPS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec peak
SR = 16000


# The reference's own published single-stream CPU figures (/root/reference/README.md:37-44: real-time factors measured with ONNX
# Runtime's CPU provider on desktop CPUs; frames/s = 31.25 / RTF in this bench's unit).  Printed inside every `cpu_baseline` beside the
# torch-CPU stand-in timed here: the stand-in is slower than ORT on a desktop part, so the published figure is the fairer yardstick.
REFERENCE_PUBLISHED = {
    "silero": {"rtf": 0.0026, "frames_per_s": 31.25 / 0.0026, "hardware": "Intel i3-12300, ONNX Runtime CPU, Ubuntu 24.04", "chunk": "512 samples", "source": "README.md:41"},
    "fsmn": {"rtf": 0.0047, "frames_per_s": 31.25 / 0.0047, "hardware": "Intel i3-12300, ONNX Runtime CPU, Ubuntu 24.04", "chunk": "512 samples", "source": "README.md:40"},
    "marblenet": {"rtf": 0.0005, "frames_per_s": 31.25 / 0.0005, "hardware": "Intel i7-1165G7, ONNX Runtime CPU, Ubuntu 24.04", "chunk": "89000 samples", "source": "README.md:42"},
    "dfsmn": {"rtf": 0.27, "frames_per_s": 31.25 / 0.27, "hardware": "Intel i7-1165G7, ONNX Runtime CPU (4 threads), Ubuntu 24.04", "chunk": "31841 samples", "source": "README.md:43"},
    "firered": {"rtf": 0.0013, "frames_per_s": 31.25 / 0.0013, "hardware": "Intel i7-1165G7, ONNX Runtime CPU, Ubuntu 24.04", "chunk": "16000 samples", "source": "README.md:44"},
}


# ------------------------------------------------------------------------------------------------ algorithmic flops
def flop_frontend_frame(n_bins, taps, n_mels=80):
    """One STFT frame as the reference computes it on non-zero window taps: cos + sin tables x taps MACs, the power
    spectrum, a dense mel projection (the kernel's banded mel issues fewer; the reference's conv1d multiplies the zero
    padding of the window too -- neither is counted)."""
    return 2 * (2 * n_bins * taps) + 3 * n_bins + 2 * n_mels * n_bins


def flop_frontend_frame_folded(n_bins, taps, n_mels=80):
    """(f32 flops, f16 flops) one frame ISSUES in the folded front-end product (csrc/frontend.hip "Folded DFT"): the symmetric part
    multiplies taps/2 mirror pairs per table row in f32 MFMAs, the residual all taps in f16 MFMAs (16x the f32 rate); power + mel as
    in flop_frontend_frame."""
    pairs = (taps + 1) // 2
    return 2 * (2 * n_bins * pairs) + 3 * n_bins + 2 * n_mels * n_bins, 2 * (2 * n_bins * taps)


PEAK_SPLIT_TFLOPS = 2500.0 / 6.0      # f32-equivalent peak of bf16 x 3 split products: the dense bf16 MFMA peak / 6 (csrc/split3.h)
PEAK_H2_TFLOPS = 2500.0 / 3.0         # f32-equivalent peak of fp16 x 2 split products: the dense fp16 MFMA peak / 3 (csrc/split2.h)
PEAK_BY_ARITH = {"f32": 157.3, "split": PEAK_SPLIT_TFLOPS, "h2": PEAK_H2_TFLOPS}
# what the chip sustains of a pipe's nominal peak when that pipe is kept busy (it clocks to its power budget; bench.py: SUSTAINED_OF_NOMINAL)
SUSTAINED_OF_NOMINAL = {"f32": 0.87, "split": 0.80, "h2": 0.86}
SUSTAINED_SOURCE = "profiles/r04_dvfs_probe.txt (f32, bf16), profiles/r05_f16x2_probe.txt (fp16)"


def _with_sustained(r, sustained):
    r["frac_of_sustained"] = r["achieved"] / (r["peak"] * sustained)
    r["sustained_of_nominal"] = sustained
    r["sustained_source"] = SUSTAINED_SOURCE
    return r


def _roof_frontend(frames, n_bins, taps, ms, tag, fold=None):
    """Roofline entry of the front-end kernel.  fold 4 (default: dense DFT product on bf16 x 3 split operands): `achieved` = the reference's
    dense arithmetic (cos + sin products; power, dense mel) over the launch time, `peak` = what that mix could do with the DFT at the
    split products' f32-equivalent peak (2500 / 6 TFLOP/s) and the mel GEMM at the f32-MFMA peak -- never 157.3 for the split part.
    fold 1 / 2 (folded f32 product): f32-MFMA flops issued + the f16 residual's flops at their f32 time equivalent (/16), against 157.3;
    `dense_equivalent_achieved` = what the reference's dense product would need."""
    if fold is None:            # callers pass the engine's ACTUAL product kind (eng.fe.fold: the front-end falls back where a kind does not apply)
        try:
            fold = int(os.environ.get("VADX_FRONTEND_FOLD", "5"))
        except ValueError:
            fold = 5
    if fold in (4, 5):
        ar = "split" if fold == 4 else "h2"
        f_dft, f_rest = 2 * (2 * n_bins * taps), 3 * n_bins + 2 * 80 * n_bins
        total = frames * (f_dft + f_rest)
        t_min = frames * (f_dft / PEAK_BY_ARITH[ar] + f_rest / PEAK_F32_MFMA_TFLOPS)          # 1e-12 s
        t_sus = frames * (f_dft / (PEAK_BY_ARITH[ar] * SUSTAINED_OF_NOMINAL[ar]) + f_rest / (PEAK_F32_MFMA_TFLOPS * SUSTAINED_OF_NOMINAL["f32"]))
        r = _roof("frontend_split_kernel", total, ms, tag, "frontend_split_kernel",
                  note=("dense DFT product as bf16 x 3 split products" if fold == 4 else "dense DFT product as fp16 x 2 split products") +
                       ": flops as the reference computes them; peak = the DFT part at the f32-equivalent peak of the split products "
                       "(2500 / 6 for bf16 x 3, 2500 / 3 for fp16 x 2 TFLOP/s) + the power / mel part at the f32-MFMA peak")
        r["peak"] = total / t_min
        r["frac"] = r["achieved"] / r["peak"]
        r["frac_of_sustained"] = r["achieved"] / (total / t_sus)
        r["sustained_source"] = SUSTAINED_SOURCE
        r["arithmetic"] = ar
        r["frontend_kind"] = fold
        r["dense_equivalent_achieved"] = r["achieved"]
        return r
    f32, f16 = flop_frontend_frame_folded(n_bins, taps)
    r = _roof("frontend_fold_kernel", frames * (f32 + f16 / 16.0), ms, tag, "frontend_fold_kernel",
              note="folded DFT product: algorithmic f32 flops of the fold (taps / 2 mirror pairs per table row; dense mel counted, the kernel's "
                   "banded mel issues fewer) + f16 residual flops / 16 (its MFMA rate is 16x); dense_equivalent = the reference's dense cos + sin product")
    r["frontend_kind"] = fold
    r["f32_flop_per_launch"], r["f16_flop_per_launch"] = frames * f32, frames * f16
    r["dense_equivalent_achieved"] = frames * flop_frontend_frame(n_bins, taps) / (ms * 1e-3) / 1e12
    return r


def flop_fsmn_frame(d=None):
    from vadx import weights
    d = dict(weights.FSMN_DIMS if d is None else d)
    D, A, L, P, K, A2, O, n = (d["input_dim"], d["input_affine_dim"], d["linear_dim"], d["proj_dim"], d["lorder"],
                               d["output_affine_dim"], d["output_dim"], d["fsmn_layers"])
    return 2 * (D * A + A * L + n * (L * P + P * K + P * L) + L * A2 + A2 * O)


def flop_firered_frame(c=None):
    from vadx import weights
    c = dict(weights.FIRERED_CFG if c is None else c)
    D, R, M, H, P = c["idim"], c["R"], c["M"], c["H"], c["P"]
    mac = D * H + H * P + R * P * (c["N1"] + c["N2"]) + (R - 1) * (P * H + H * P) + P * H + (M - 1) * H * H + H * c["odim"]
    return 2 * mac


def flop_marblenet_out_frame():
    """per 20 ms output frame (after the stride-2 first block): depthwise + pointwise + residual 1x1 + decoder"""
    from vadx import weights
    mac, cin = 0, 80
    for filt, rep, k, _s, _d, residual, sep in weights.MARBLENET_BLOCKS:
        block_cin = cin
        for _ in range(rep):
            mac += (cin * k if sep else 0) + cin * filt
            cin = filt
        if residual:
            mac += block_cin * filt
    return 2 * (mac + 2 * cin)


def flop_marblenet_h2_out_frame(eng=None):
    """the part of flop_marblenet_out_frame() that runs as fp16 x 2 split products when the engine's mode is "h2": EVERY 1x1 conv of the
    published layout -- the prologue's point-wise conv, the point-wise + residual convs of the three fused blocks (kgemm_h / pgemm) and both
    tail blocks (qgemm_group on planes); only the depthwise filters and the two-class decoder stay on the float32 pipes.  With an engine the
    figure is read off the stage list its launches are fed from (MarbleNetEngine.h2_macs_per_out_frame: the stages that carry `_h`
    fragments), so the two cannot drift (the prologue and tail, 44 % of these MACs, used to be priced at the f32 peak)."""
    if eng is not None:
        return 2 * eng.h2_macs_per_out_frame()
    from vadx import weights
    mac, cin = 0, 80
    for filt, rep, _k, _s, _d, residual, _sep in weights.MARBLENET_BLOCKS:
        block_cin = cin
        for _ in range(rep):
            mac += cin * filt
            cin = filt
        if residual:
            mac += block_cin * filt
    return 2 * mac


def _roof_mix(kernel, flop, flop_h2, ms, tag=None, kernel_substr=None, note=None):
    """A group of launches whose products run partly on fp16 x 2 split products and partly on f32 MFMAs, priced against the peak of ITS pipe
    mix: peak = flops / (the time the two pipes need at their own peaks)"""
    f_f32 = max(flop - flop_h2, 0.0)
    t_min = flop_h2 / PEAK_BY_ARITH["h2"] + f_f32 / PEAK_BY_ARITH["f32"]
    t_sus = flop_h2 / (PEAK_BY_ARITH["h2"] * SUSTAINED_OF_NOMINAL["h2"]) + f_f32 / (PEAK_BY_ARITH["f32"] * SUSTAINED_OF_NOMINAL["f32"])
    r = _roof(kernel, flop, ms, tag, kernel_substr, note, split="f32")
    r.update({"arithmetic": "mix" if flop_h2 else "f32", "peak": flop / t_min, "frac": r["achieved"] / (flop / t_min),
              "frac_of_sustained": r["achieved"] / (flop / t_sus), "flop_on_split_products": flop_h2})
    r.pop("sustained_of_nominal", None)
    return r


def flop_dfsmn_window(T=101, TA=51, F=160, ch=20):
    """One 16001-sample near+far window (Export_DFSMN_VAD.py:317-354), by stage."""
    def lstm(i, h, bi=False, layers=1):
        m, d = 0, i
        for _ in range(layers):
            m += 4 * h * (d + h) * (2 if bi else 1)
            d = h * (2 if bi else 1)
        return m
    out = {}
    out["lstm_f"] = 2 * T * (F * (lstm(4, ch, True) + 2 * ch * ch) + 10 * 81 * (lstm(2 * ch, ch, True) + 2 * ch * 2 * ch))
    out["dft_f"] = 2 * T * 10 * ch * (2 * 81 * F + 162 * F)                              # forward (cos, sin) + pinv inverse
    pw = F * ((4 + ch) * ch + 3 * ch * 2)                                                # in_conv, out_conv
    for cin in (ch,) * 6 + (2 * ch,) * 4:
        pw += F * (2 * cin * ch + 3 * ch * ch)                                           # gate, input, (3,1) conv
    pw += F * (2 * ch * ch + ch * 2 * ch)                                                # the two time-LSTM output linears
    out["pw_conv"] = 2 * T * pw
    out["lstm_t"] = 2 * T * F * (lstm(ch, 2 * ch, False, 2) + lstm(2 * ch, ch))
    # the fused block kernels (csrc/dfsmn_cfb.hip) regroup the same arithmetic: cfb_front = gate / input / (3,1) convs + forward DFT,
    # cfb_back = CepsUnit Linear + pinv inverse DFT (NOT added to the total: they are pw_conv / dft_f / lstm_f's linear, regrouped)
    conv_cfb = sum(F * (2 * cin * ch + 3 * ch * ch) for cin in (ch,) * 6 + (2 * ch,) * 4)
    regroup = {"cfb_front": 2 * T * (conv_cfb + 10 * ch * 2 * 81 * F), "cfb_back": 2 * T * 10 * (81 * 2 * ch * 2 * ch + ch * 162 * F)}
    out["istft"] = 2 * T * 320 * 319
    out["frontend"] = 2 * (2 * T * 2 * F * 319) + 3 * TA * (flop_frontend_frame(513, 640))
    from vadx import weights
    m = weights.DFSMN_MASK
    H, H2 = m["hidden"], m["fsmn_hidden"]
    out["mask_net"] = 2 * TA * (240 * H + m["layers"] * (H * H2 + H2 * H + H * m["lorder"]) + H)
    out["total"] = sum(out.values())
    out.update(regroup)
    return out


def flop_dfsmn_by_entry(fused, T=101, TA=51, F=160, ch=20):
    """The same arithmetic attributed to the C-ABI entry point that EXECUTES it (flop_dfsmn_window groups it by the reference's
    stages).  fused: the gated blocks run as cfb_front -> lstm_f -> cfb_back, so the gate / input / (3,1) convs and the forward DFT
    belong to cfb_front, the CepsUnit linear and the inverse DFT to cfb_back, and the pw_conv launches that remain are in_ch_lstm's
    linear, in_conv and out_conv (round 3 priced those three launches with all of pw_conv's flops: frac 2.46).  The two time-LSTM
    output linears run inside the lstm_t kernels either way.  Sums to flop_dfsmn_window()['total'] minus front-end / istft / mask-net."""
    st = flop_dfsmn_window(T, TA, F, ch)
    lin_t = 2 * T * F * (2 * ch * ch + ch * 2 * ch)
    lin_in = 2 * T * F * 2 * ch * ch                                           # in_ch_lstm.linear (40 -> 20): a pw_conv launch
    lin_ceps = 2 * T * 10 * 81 * 2 * ch * 2 * ch                               # CepsUnit linear (40 -> 40 on 81 bins)
    e = {"lstm_t": st["lstm_t"] + lin_t, "lstm_f": st["lstm_f"] - lin_in - lin_ceps}
    if fused:
        e["cfb_front"], e["cfb_back"] = st["cfb_front"], st["cfb_back"]
        e["pw_conv"] = lin_in + 2 * T * F * ((4 + ch) * ch + 3 * ch * 2)
        e["dft_f"] = 0
    else:
        e["pw_conv"] = st["pw_conv"] - lin_t + lin_in + lin_ceps
        e["dft_f"] = st["dft_f"]
        e["cfb_front"] = e["cfb_back"] = 0
    return e


# ------------------------------------------------------------------------------------------------ helpers
def synth_pcm16(torch, device, batch, samples, seed, loud=3000.0, quiet=30.0):
    """int16 burst clips generated on the GPU (every clip unique): 0.5-2 s segments alternating N(0,loud) / N(0,quiet)
    (SURVEY 8(d) recipe; the host-side twin is vadx.weights.burst_clips)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((batch, samples), dtype=torch.int16, device=device)
    chunk = 512
    nseg = int(samples / (0.5 * SR)) + 2
    for b0 in range(0, batch, chunk):
        nb = min(chunk, batch - b0)
        dur = (torch.rand((nb, nseg), generator=g, device=device) * 1.5 + 0.5) * float(SR)
        edges = torch.cumsum(dur, dim=1)
        pos = torch.arange(samples, device=device, dtype=torch.float32).unsqueeze(0).expand(nb, -1).contiguous()
        seg = torch.searchsorted(edges, pos)
        first = torch.randint(0, 2, (nb, 1), generator=g, device=device)
        isloud = ((seg + first) % 2) == 0
        sigma = torch.where(isloud, torch.tensor(loud, device=device), torch.tensor(quiet, device=device))
        x = torch.randn((nb, samples), generator=g, device=device) * sigma
        out[b0:b0 + nb] = torch.clamp(torch.round(x), -32768, 32767).to(torch.int16)
        del dur, edges, pos, seg, isloud, sigma, x
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_cpu(fn, units_per_call, budget_s, threads=(1, 4, 16)):
    """units/s of `fn` (one batch-1 call of the oracle) on the host: a short calibration picks the intra-op thread count,
    then calls are repeated for ~budget_s (checked after every call)."""
    import torch
    ncpu = os.cpu_count() or 1
    best_thr, best = 1, 0.0
    with torch.no_grad():
        for thr in threads:
            if thr > ncpu:
                continue
            torch.set_num_threads(thr)
            fn()
            t0, n = time.perf_counter(), 0
            while n < 1 or time.perf_counter() - t0 < 0.25:
                fn()
                n += 1
            r = n / (time.perf_counter() - t0)
            if r > best:
                best_thr, best = thr, r
        torch.set_num_threads(best_thr)
        t0, n = time.perf_counter(), 0
        while n < 1 or time.perf_counter() - t0 < budget_s:
            fn()
            n += 1
        el = time.perf_counter() - t0
    return n * units_per_call / el, best_thr, n, el


def device_ms(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


# FETCH_SIZE calibration (MI355X_MICROARCH.md 'HBM': gfx950 tallies a 128-B request at 64 B -- "double it" holds for wide coalesced
# reads only, other access widths are to be calibrated on a known byte count).  Calibrated here: the DFSMN frequency-axis LSTM
# reads its input as 64-B rows 5 KB apart, and its raw FETCH_SIZE equals the tensor's bytes (1.302 M KiB counted for 1.361 M KiB
# read, tools/prof_lstmf_fetch.sh; what-if builds without loads / without stores separate the two directions) -- factor 1.  The
# time-axis LSTMs read 4-B / 16-B pieces of 64-B rows the same way.
FETCH_FACTOR_64B = ("lstm_f", "lstm_t")


def profiled_kernel_traffic(tag, kernel_substr):
    """HBM bytes PER PASS of every kernel whose name contains `kernel_substr`, from the newest committed rocprofv3 PMC summary
    profiles/r*_{tag}/SUMMARY.txt (tools/profile_secondary.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes of the BASELINE-size
    workload; lines `<kernel> calls_per_pass=<n> FETCH_SIZE sum_per_pass=<KiB>`).  FETCH is doubled for kernels that read wide
    coalesced runs and taken as counted for the 64-B-row readers (FETCH_FACTOR_64B); both counters are KiB."""
    best = None
    factor = 1.0 if any(k in kernel_substr for k in FETCH_FACTOR_64B) else 2.0
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}", "SUMMARY.txt"))):
        fetch = write = 0.0
        calls = 0
        seen = set()
        for line in open(path):
            if kernel_substr not in line or "sum_per_pass=" not in line:
                continue
            name = line.split("calls_per_pass=")[0].strip()
            val = float(line.split("sum_per_pass=")[1])
            if " FETCH_SIZE " in line:
                fetch += val
                if name not in seen:
                    seen.add(name)
                    calls += int(line.split("calls_per_pass=")[1].split()[0])
            elif " WRITE_SIZE " in line:
                write += val
        if fetch > 0 or write > 0:
            best = {"bytes": (factor * fetch + write) * 1024.0, "source": os.path.relpath(path, ROOT), "calls_per_pass": calls,
                    "fetch_factor": factor}
    return best


def profiled_pass_traffic(tag):
    """HBM bytes of ALL launches of one pass of a secondary workload, from the newest profiles/r*_{tag}/SUMMARY.txt: every
    kernel's FETCH_SIZE (doubled, except the 64-B-row readers of FETCH_FACTOR_64B) + WRITE_SIZE."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}", "SUMMARY.txt"))):
        total, launches, seen = 0.0, 0, set()
        for line in open(path):
            if "sum_per_pass=" not in line or line.startswith("#"):
                continue
            name = line.split("calls_per_pass=")[0].strip()
            val = float(line.split("sum_per_pass=")[1])
            if " FETCH_SIZE " in line:
                total += (1.0 if any(k in name for k in FETCH_FACTOR_64B) else 2.0) * val * 1024.0
                if name not in seen:
                    seen.add(name)
                    launches += int(line.split("calls_per_pass=")[1].split()[0])
            elif " WRITE_SIZE " in line:
                total += val * 1024.0
        if total > 0:
            best = {"bytes": total, "launches_per_pass": launches, "source": os.path.relpath(path, ROOT)}
    return best


def _gemm_arith(eng=None):
    """the arithmetic an engine's dense layers actually run ("f32" | "split" | "h2"): its ArithBlobs' mode (which falls back where a mode
    cannot be packed), else the module default"""
    from vadx import _lib
    if eng is not None and hasattr(eng, "blobs"):
        return eng.blobs.mode()
    return _lib.gemm_mode()


def _roof(kernel, flop, ms, tag=None, kernel_substr=None, note=None, split=False):
    """split: False / "f32" = f32 MFMAs (peak 157.3); True / "split" = bf16 x 3 split products, priced against THEIR f32-equivalent peak
    (2500 / 6); "h2" = fp16 x 2 split products (2500 / 3) -- never against 157.3"""
    ar = {False: "f32", True: "split"}.get(split, split)
    ach = flop / (ms * 1e-3) / 1e12
    tr = profiled_kernel_traffic(tag, kernel_substr) if tag else None
    peak = PEAK_BY_ARITH[ar]
    r = {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "arithmetic": ar,
         "frac": ach / peak, "frac_of_sustained": ach / (peak * SUSTAINED_OF_NOMINAL[ar]), "sustained_of_nominal": SUSTAINED_OF_NOMINAL[ar],
         "flop_per_launch": flop, "ms": ms,
         "traffic": tr["bytes"] if tr else None, "traffic_unit": "B per pass (all launches of the kernel)",
         "traffic_source": tr["source"] if tr else None,
         "traffic_GBps": (tr["bytes"] / (ms * 1e-3) / 1e9) if tr else None}
    if note:
        r["note"] = note
    return r


LSTM_T_H2_SHARE = 23200.0 / 28800.0     # of vadx_dfsmn_lstm_t's flops: the two-layer net (fp16 x 2 form) against the one-layer net (f32 MFMAs)


def _whole_pass_roof(nwin, flop_total, fle, groups, on_split, ms, f_h2=0.0):
    """The DFSMN pass against the peak of ITS pipe mix (as bench.encoder_roofline does for one kernel): the flops of the entry points that
    run bf16 x 3 split products at 2500 / 6, those on fp16 x 2 (f_h2) at 2500 / 3, everything else at 157.3 -- peak = total flops / the time
    the pipes need at their own peaks."""
    f_split = sum(nwin * fle[k] for k in groups if k in on_split and groups[k] > 0)
    f_all = nwin * flop_total
    f_f32 = max(f_all - f_split - f_h2, 0.0)
    t_min = f_split / PEAK_SPLIT_TFLOPS + f_h2 / PEAK_H2_TFLOPS + f_f32 / PEAK_F32_MFMA_TFLOPS
    t_sus = (f_split / (PEAK_SPLIT_TFLOPS * SUSTAINED_OF_NOMINAL["split"]) + f_h2 / (PEAK_H2_TFLOPS * SUSTAINED_OF_NOMINAL["h2"])
             + f_f32 / (PEAK_F32_MFMA_TFLOPS * SUSTAINED_OF_NOMINAL["f32"]))
    ach = f_all / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "all DFSMN launches", "achieved": ach, "peak": f_all / t_min, "unit": "TFLOP/s", "frac": ach / (f_all / t_min),
            "frac_of_sustained": ach / (f_all / t_sus), "sustained_source": SUSTAINED_SOURCE, "arithmetic": "mix" if f_split else "f32",
            "flop_per_launch": f_all, "flop_on_split_products": f_split, "flop_on_fp16x2_products": f_h2, "ms": ms,
            "note": "peak = the pass's pipe mix: flops of the bf16 x 3 entry points at 2500 / 6 TFLOP/s, of the fp16 x 2 time LSTM at 2500 / 3, the rest "
                    "at the f32-MFMA peak"}


def bytes_dfsmn_pw_window(T=101, F=160, ch=20):
    """HBM bytes one window's pw_conv launches must move: every launch reads its input tensors and writes its outputs once
    (FT layout: tiles of 16 frames; U = one 20-channel tensor of a window).  The (3,1) / 1x1 convs have ~100 flop per 64-byte
    row: these launches are HBM-bound, not matrix-pipe-bound."""
    U = ((T + 15) // 16) * ch * F * 16 * 4
    Uc = ((T + 15) // 16) * 2 * ch * 81 * 16 * 4               # a 40-channel x 81-bin CepsUnit tensor
    n = 3 * U + (1 + 4 / ch + 1) * U                            # in_ch_lstm linear (40 -> 20), in_conv (24 -> 20)
    for cin in (ch,) * 6 + (2 * ch,) * 4:
        n += (cin / ch + 2) * U + 2 * Uc + 3 * U                # CFB front (in -> gx, r), ceps linear, CFB back (gx, ceps -> out)
    n += (3 + 2 / ch) * U                                       # out_conv (60 -> 2)
    return int(n)


def _roof_hbm(kernel, algorithmic_bytes, ms, tag=None, kernel_substr=None, note=None):
    ach = algorithmic_bytes / (ms * 1e-3) / 1e9
    tr = profiled_kernel_traffic(tag, kernel_substr) if tag else None
    r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBPS,
         "algorithmic_bytes_per_pass": algorithmic_bytes, "ms": ms, "traffic": tr["bytes"] if tr else None,
         "traffic_unit": "B per pass (all launches of the kernel)", "traffic_source": tr["source"] if tr else None,
         "traffic_ratio": (tr["bytes"] / algorithmic_bytes) if tr else None}
    if note:
        r["note"] = note
    return r


def _hbm(bytes_algorithmic, ms, tag=None):
    """SURVEY 8(d) view of a whole pass: algorithmic bytes (PCM in + scores out) / time against the HBM peak, and -- when a PMC
    profile of the full-size pass is committed -- what ALL its launches really moved over those bytes (`traffic_ratio`)."""
    g = bytes_algorithmic / (ms * 1e-3) / 1e9
    out = {"algorithmic_bytes": bytes_algorithmic, "achieved_GBps": g, "peak_GBps": PEAK_HBM_GBPS, "frac": g / PEAK_HBM_GBPS}
    tr = profiled_pass_traffic(tag) if tag else None
    if tr:
        out.update({"traffic_bytes_per_pass": tr["bytes"], "traffic_ratio": tr["bytes"] / bytes_algorithmic,
                    "launches_per_pass": tr["launches_per_pass"], "traffic_source": tr["source"]})
    return out


def _trace(fn):
    from vadx import _lib
    import torch
    torch.cuda.synchronize()
    with _lib.trace() as tr:
        fn()
    return {k: round(v, 4) for k, v in sorted(tr.ms.items(), key=lambda kv: -kv[1])}, tr.calls


# ------------------------------------------------------------------------------------------------ CPU baselines
# (the ONLY functions of this file that touch oracle/: the torch-CPU restatement timed on the host, batch 1, one call per window,
#  exactly how the reference drives its ORT session -- a reported, non-target baseline)
def _cpu_entry(rate, thr_n, sample, model):
    return {"value": rate, "unit": "frames/s", "cores": thr_n, "kind": "port", "cpu": cpu_model(), "sample": sample,
            "reference_published": REFERENCE_PUBLISHED[model]}


def cpu_baseline_fsmn(budget_s, stride):
    import torch
    from oracle import fsmn as ofs
    from vadx import weights
    w = {k: torch.from_numpy(v) for k, v in weights.fsmn_synthetic(1234).items()}
    fe = ofs.Frontend(16000)
    a = torch.from_numpy(weights.burst_clips(1, 16000, seed=3)).reshape(1, 1, -1)
    caches = [torch.zeros(1, 128, 19, 1) for _ in range(4)]
    thr, nz = torch.tensor([1.0]), torch.tensor([4.0])
    rate, thr_n, calls, el = timed_cpu(lambda: ofs.forward(fe, w, a, caches, thr, nz), stride / 512.0, budget_s)
    return _cpu_entry(rate, thr_n, f"{calls} one-second windows, batch 1, one oracle call per window at stride {stride} "
                                   f"(torch-CPU stand-in for ORT-CPU), {el:.1f} s", "fsmn")


def cpu_baseline_marblenet(budget_s, n):
    import torch
    from oracle import marblenet as omb
    from vadx import weights
    w = {k: torch.from_numpy(v) for k, v in weights.marblenet_synthetic(1234).items()}
    fe = omb.Frontend()
    a = torch.from_numpy(weights.burst_clips(1, n, seed=4)).reshape(1, 1, -1)
    rate, thr_n, ncall, el = timed_cpu(lambda: omb.forward(fe, w, a), n / 512.0, budget_s)
    return _cpu_entry(rate, thr_n, f"{ncall} clips of {n} samples, batch 1, one oracle call per clip (torch-CPU stand-in for "
                                   f"ORT-CPU), {el:.1f} s", "marblenet")


def cpu_baseline_firered(budget_s):
    import torch
    from oracle import firered as ofr
    from vadx import weights
    w = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in weights.firered_synthetic(1234).items()}
    fe = ofr.Frontend()
    a = torch.from_numpy(weights.burst_clips(1, 16000, seed=5)).reshape(1, 1, -1)
    rate, thr_n, ncall, el = timed_cpu(lambda: ofr.forward(fe, w, a), 16000 / 512.0, budget_s)
    return _cpu_entry(rate, thr_n, f"{ncall} one-second windows, batch 1, one oracle call per window (torch-CPU stand-in for "
                                   f"ORT-CPU), {el:.1f} s", "firered")


def cpu_baseline_dfsmn(budget_s, stride):
    import torch
    from oracle import dfsmn as od
    from vadx import weights
    w = {k: torch.from_numpy(v) for k, v in weights.dfsmn_synthetic(1234).items()}
    w["mask.shift"] = w["mask.shift"] + torch.log(torch.tensor(32768.0 ** 2))
    fe = od.Frontend()
    a = torch.from_numpy(weights.burst_clips(1, 16001, seed=6)).reshape(1, 1, -1)
    b = torch.from_numpy(weights.burst_clips(1, 16001, seed=7)).reshape(1, 1, -1)
    nf = weights.DFSMN_MASK["layers"]
    rate, thr_n, ncall, el = timed_cpu(lambda: od.forward(fe, w, a, b, nf), stride / 512.0, budget_s, threads=(4,))
    return _cpu_entry(rate, thr_n, f"{ncall} windows of 16001 samples, batch 1, one oracle call per window at stride {stride} "
                                   f"(torch-CPU stand-in for ORT-CPU; the reference pins 4 ORT threads for this model), {el:.1f} s", "dfsmn")


# ------------------------------------------------------------------------------------------------ the workloads
def fsmn_c3(torch, device, reps, cpu, clips=4096, log=lambda m: None):
    from vadx import fsmn, weights
    eng = fsmn.FsmnEngine(weights.fsmn_synthetic(1234), device=device)
    lb, stride = eng.grid()
    n = 160000
    W = -(-(n - eng.L) // stride) + 1                       # Inference_FSMN_VAD_ONNX.py:88-92 window grid: 15
    padded = (W - 1) * stride + eng.L
    audio = synth_pcm16(torch, device, clips, padded, seed=1303)
    log(f"fsmn: {clips} x {padded} int16 resident, {W} windows/clip")
    run = lambda: eng.flags(audio, W)                       # noqa: E731
    ms = device_ms(torch, run, reps)
    split, _ = _trace(run)
    frames10 = clips * W * eng.T
    net_ms = split.get("vadx_fsmn_clips", ms)
    out = {"workload": f"FSMN-VAD f32, batch={clips} synthetic 10 s clips -> silence flags (window grid {W} x 1 s at stride "
                       f"{stride}, noise-floor feedback and look-ahead vote on device)",
           "clips": clips, "samples_per_clip": n, "windows_per_clip": W, "ms": ms,
           "frames_per_s": clips * n / 512 / (ms * 1e-3), "kernel_ms": split,
           "roofline": _roof("fsmn_clips_kernel", frames10 * flop_fsmn_frame(), net_ms, "fsmn", "fsmn_clips_kernel", split=_gemm_arith(eng)),
           "roofline_frontend": _roof_frontend(frames10, 257, 400, split.get("vadx_frontend_logmel_means", split.get("vadx_frontend_logmel", ms)), "fsmn",
                                               fold=eng.fe.fold),
           "range_fallbacks": eng.blobs.range_fallbacks,
           "hbm": _hbm(clips * (padded * 2 + (W * (eng.T - lb) + lb)), ms, "fsmn"), "cpu_baseline": None}
    # NOT the default path, reported beside it: the opt-in time x frequency fold of the front-end (VADX_FRONTEND_FOLD=3: a quarter of the
    # dense MACs, noisier on bands far below a frame's peak -- DESIGN 4c); the entry counts the silence flags that differ from the default path's on this batch
    prev = os.environ.get("VADX_FRONTEND_FOLD")
    try:
        os.environ["VADX_FRONTEND_FOLD"] = "3"
        eng3 = fsmn.FsmnEngine(weights.fsmn_synthetic(1234), device=device)
        run3 = lambda: eng3.flags(audio, W)                 # noqa: E731
        f3, f2 = run3(), run()
        ndiff, ntot = int((f3 != f2).sum().item()), int(f2.numel())
        ms3 = device_ms(torch, run3, reps)
        split3, _ = _trace(run3)
        out["opt_in_frontend_kind3"] = {"fold": int(eng3.fe.fold), "ms": ms3, "frames_per_s": clips * n / 512 / (ms3 * 1e-3),
                                        "frontend_ms": split3.get("vadx_frontend_logmel_means", split3.get("vadx_frontend_logmel")), "flags_differing_from_default": ndiff, "flags_total": ntot}
        del eng3
    except Exception as e:                                  # noqa: BLE001
        out["opt_in_frontend_kind3"] = {"error": f"{type(e).__name__}: {e}"}
    finally:
        if prev is None:
            os.environ.pop("VADX_FRONTEND_FOLD", None)
        else:
            os.environ["VADX_FRONTEND_FOLD"] = prev
    del audio
    if cpu:
        out["cpu_baseline"] = cpu_baseline_fsmn(cpu, stride)
    return out


def marblenet_c4(torch, device, reps, cpu, clips=8192, log=lambda m: None, tag="marblenet"):
    from vadx import marblenet, weights
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234), device=device)
    n = 89431
    audio = synth_pcm16(torch, device, clips, n, seed=1404)
    log(f"marblenet: {clips} x {n} int16 resident")
    run = lambda: eng.run(audio)                            # noqa: E731
    ms = device_ms(torch, run, reps)
    split, calls = _trace(run)
    T = n // 160 + 1
    Tout = (T + 2 * 5 - 10 - 1) // 2 + 1
    fe_ms = split.get("vadx_frontend_logmel", ms)
    net_ms = sum(v for k, v in split.items() if k != "vadx_frontend_logmel")   # every launch after the front end
    out = {"workload": f"NVIDIA Frame-VAD MarbleNet v2.0 f32, batch={clips} clips of {n} samples (one dynamic-axis window each) "
                       "-> per-20-ms speech probabilities",
           "clips": clips, "samples_per_clip": n, "ms": ms, "frames_per_s": clips * n / 512 / (ms * 1e-3),
           "kernel_ms": split, "kernel_calls": calls,
           "roofline": _roof_frontend(clips * T, 257, 400, fe_ms, tag, fold=eng.frontend(n).fold),
           "roofline_net": _roof_mix("marblenet encoder+decoder launches", clips * Tout * flop_marblenet_out_frame(),
                                     clips * Tout * flop_marblenet_h2_out_frame(eng) if eng.mode() == "h2" else 0.0, net_ms, tag, "vadx::marblenet::",
                                     note="sum of the encoder / classifier entries in kernel_ms (prologue, fused block pairs, tail); every 1x1 conv "
                                          "runs on fp16 x 2 split products (priced at 2500 / 3), the depthwise filters and the decoder on the f32 pipes"),
           "encoder_arithmetic": eng.mode(), "range_fallbacks": eng.range_fallbacks,
           "hbm": _hbm(clips * (n * 2 + 2 * Tout * 4), ms, tag), "cpu_baseline": None}
    del audio
    if cpu:
        out["cpu_baseline"] = cpu_baseline_marblenet(cpu, n)
    return out


def firered_c5(torch, device, reps, cpu, clips=2048, log=lambda m: None):
    from vadx import firered, weights
    eng = firered.FireRedEngine(weights.firered_synthetic(1234), device=device)
    n, W = 160000, 10
    audio = synth_pcm16(torch, device, clips, n, seed=1505)
    log(f"firered: {clips} x {n} int16 resident")
    run = lambda: eng.run(audio, W)                         # noqa: E731
    ms = device_ms(torch, run, reps)
    split, _ = _trace(run)
    frames10 = clips * W * 98
    out = {"workload": f"FireRedVAD f32, batch={clips} synthetic 10 s clips ({W} x 1 s windows, 98 frames each) -> per-10-ms "
                       "speech probabilities",
           "clips": clips, "samples_per_clip": n, "ms": ms, "frames_per_s": clips * n / 512 / (ms * 1e-3), "kernel_ms": split,
           "roofline": _roof("firered_kernel", frames10 * flop_firered_frame(), split.get("vadx_firered_run", ms), "firered",
                             "firered_kernel", split=_gemm_arith(eng)),
           "roofline_frontend": _roof_frontend(frames10, 201, 400, split.get("vadx_frontend_logmel", ms), "firered", fold=eng.fe.fold),
           "range_fallbacks": eng.blobs.range_fallbacks,
           "hbm": _hbm(clips * (n * 2 + W * 98 * 4), ms, "firered"), "cpu_baseline": None}
    del audio
    if cpu:
        out["cpu_baseline"] = cpu_baseline_firered(cpu)
    return out


def dfsmn_c5(torch, device, reps, cpu, clips=2048, log=lambda m: None, sub_batch=3072):
    from vadx import dfsmn, weights
    eng = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), device=device, sub_batch=sub_batch)
    lb, stride = eng.grid()
    n = 160000
    W = -(-(n - eng.L) // stride) + 1                       # 15 windows of 16001 samples at stride 10881
    padded = (W - 1) * stride + eng.L
    near = synth_pcm16(torch, device, clips, padded, seed=1606)
    far = synth_pcm16(torch, device, clips, padded, seed=1607)
    log(f"dfsmn: 2 x {clips} x {padded} int16 resident, {W} windows/clip")
    run = lambda: eng.run(near, far, W, stride)             # noqa: E731
    ms = device_ms(torch, run, max(1, reps - 1))
    split, calls = _trace(run)
    fl = flop_dfsmn_window()
    nwin = clips * W
    fle = flop_dfsmn_by_entry(fused=split.get("vadx_dfsmn_cfb_front", 0.0) > 0)
    groups = {"lstm_f": split.get("vadx_dfsmn_lstm_f", 0.0), "dft_f": split.get("vadx_dfsmn_dft_f", 0.0),
              "pw_conv": split.get("vadx_dfsmn_pw_conv", 0.0), "lstm_t": split.get("vadx_dfsmn_lstm_t", 0.0) + split.get("vadx_dfsmn_lstm_t_ex", 0.0),
              "cfb_front": split.get("vadx_dfsmn_cfb_front", 0.0), "cfb_back": split.get("vadx_dfsmn_cfb_back", 0.0)}
    # entries whose matrix products run as bf16 x 3 split products (these kernels have no fp16 x 2 form yet: "h2" maps to bf16 x 3 there):
    # priced against the f32-equivalent peak of that pipe (cfb_back's split form is opt-in, VADX_CFB_BACK=split: by default it runs f32 MFMAs)
    on_split = ({"lstm_f", "cfb_front"} | ({"cfb_back"} if os.environ.get("VADX_CFB_BACK") == "split" else set())) if _gemm_arith() != "f32" else set()
    h2 = _gemm_arith() == "h2"                               # the two LSTMs have fp16 x 2 forms (the CepsUnit's lstm_f, the two-layer lstm_t)
    if h2:
        on_split = on_split - {"lstm_f"}
    f_h2_lstm_t = nwin * fle["lstm_t"] * LSTM_T_H2_SHARE if h2 else 0.0
    by_entry = {k: _roof(f"vadx_dfsmn_{k}", nwin * fle[k], v, "dfsmn", k, split=("h2" if (h2 and k == "lstm_f") else k in on_split))
                for k, v in groups.items() if v > 0 and k != "lstm_t"}
    if groups["lstm_t"] > 0:
        by_entry["lstm_t"] = _roof_mix("vadx_dfsmn_lstm_t", nwin * fle["lstm_t"], f_h2_lstm_t, groups["lstm_t"], "dfsmn", "lstm_t",
                                       note="two launches per pass: the two-layer net (fp16 x 2 split products) and the one-layer net (f32 MFMAs)")
    dom = max(groups, key=groups.get)
    if dom == "pw_conv":                                     # (only the unfused chain is dominated by the HBM-bound pw_conv launches)
        dom = max((k for k in groups if k != "pw_conv"), key=groups.get)
    out = {"workload": f"DFSMN near+far f32 (SDAEC ICCRN echo canceller + mask-net VAD), batch={clips} clip pairs of 10 s = "
                       f"{nwin} windows of 16001 samples, measured at full size in sub-batches of {sub_batch} windows",
           "clip_pairs": clips, "samples_per_clip": n, "windows": nwin, "ms": ms,
           "frames_per_s": clips * n / 512 / (ms * 1e-3), "kernel_ms": split, "kernel_calls": calls,
           "flop_per_window": fl,
           "roofline": _roof(f"{dom} launches (vadx_dfsmn_{dom})", nwin * fle[dom], groups[dom], "dfsmn", dom,
                             split=("h2" if (h2 and dom == "lstm_f") else dom in on_split),
                             note="all launches of the entry point that takes the most time; flops as the reference computes them "
                                  "(the kernel issues more: 20 output channels pad to 32 MFMA rows)"),
           "roofline_by_entry": by_entry,
           "roofline_whole_pass": _whole_pass_roof(nwin, fl["total"], fle, groups, on_split, ms, f_h2_lstm_t + (nwin * fle["lstm_f"] if h2 else 0.0)),
           "range_fallbacks": eng.range_fallbacks,
           "hbm": _hbm(clips * (2 * padded * 2 + W * eng.T_A * 4), ms, "dfsmn"), "cpu_baseline": None}
    del near, far
    if cpu:
        out["cpu_baseline"] = cpu_baseline_dfsmn(cpu, stride)
    return out


def marblenet_c4_sharded(torch, device, dist, rank, world, reps=3, clips=8192, log=lambda m: None, feed=True):
    """BASELINE config 4 as the config states it: 8192 clips of 89 431 samples STRONG-sharded over the node's GPUs -- rank r owns
    clips shard_bounds(8192, r, world), no data-path collective; the time is the max over ranks between two barriers, the value
    the whole job's 512-hop frames / s.  `feed`: the same shard timed from PINNED HOST int16 (vadx.feed.HostPcmFeed) -- what the
    node sustains from host memory.  Every rank returns the same dict."""
    from vadx import feed as vfeed, marblenet, shard, weights
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234), device=device)
    n = 89431
    lo, hi = shard.shard_bounds(clips, rank, world)
    audio = synth_pcm16(torch, device, hi - lo, n, seed=1404 + rank)
    fence = lambda: shard.fence(dist, torch.cuda.synchronize)           # noqa: E731

    def timed(fn):
        fn()
        best = None
        for _ in range(reps):
            fence()
            t0 = time.perf_counter()
            fn()
            fence()
            el = shard.max_over_ranks(dist, time.perf_counter() - t0, device)
            best = el if best is None else min(best, el)
        return best

    el = timed(lambda: eng.run(audio))
    out = {"workload": f"MarbleNet v2.0 f32, {clips} clips x {n} samples strong-sharded over {world} GPU(s), no collective",
           "clips": clips, "n_gpus": world, "shard": [lo, hi], "ms": el * 1e3, "frames_per_s": clips * n / 512 / el, "scaling": "strong"}
    if feed:
        # Everything that can fail on ONE rank (pinned allocation, the feed's buffers, the first runs) happens locally first; the ranks
        # then AGREE (all_ok) before any of them enters timed(), whose barriers / all-reduces every rank must reach.
        err, host, f, same = None, None, None, None
        try:
            host = vfeed.pin(audio.cpu())
            f = vfeed.HostPcmFeed(device, n, max(64, min(256, (hi - lo) // 8)))      # at least eight chunks per shard; tools/feed_sweep.py, 8192 clips: 64 clips 40.9 ms, 128 31.3, 256 27.0, 512 27.3, 1024 28.4 (upload alone 25.5)
            ref = eng.run(audio)
            got = eng.run_from_host(host, feed=f)
            same = bool(torch.equal(got[1], ref[1]))
        except Exception as e:                               # noqa: BLE001  (pinned allocation can be refused)
            err = f"{type(e).__name__}: {e}"
        if shard.all_ok(dist, err is None, device):
            elf = timed(lambda: eng.run_from_host(host, feed=f))     # a failure in here is fatal for the job on every rank: not caught
            out["feed"] = {"ms": elf * 1e3, "frames_per_s": clips * n / 512 / elf, "upload_bytes_per_gpu": (hi - lo) * n * 2,
                           "scores_bit_identical_to_resident": same}
        else:
            out["feed"] = {"error": err or "another rank could not set its host feed up"}
    log(f"marblenet_c4 sharded x{world}: {out['ms']:.2f} ms")
    return out


def compact(entry):
    """The few numbers of one workload that go into bench.py's JSON line (the full entry goes to the detail file)."""
    if entry is None or "error" in entry:
        return entry
    ro = entry.get("roofline") or {}
    cpu = entry.get("cpu_baseline") or {}
    hbm = entry.get("hbm") or {}
    r3 = lambda v: None if v is None else float(f"{v:.4g}")      # noqa: E731
    out = {"ms": r3(entry.get("ms")), "frames_per_s": r3(entry.get("frames_per_s")), "kernel": ro.get("kernel"), "frac": r3(ro.get("frac")),
           "bound": ro.get("bound"), "traffic_ratio": r3(hbm.get("traffic_ratio")), "launches": hbm.get("launches_per_pass"),
           "cpu": r3(cpu.get("value")), "cpu_published": r3((cpu.get("reference_published") or {}).get("frames_per_s"))}
    return {k: v for k, v in out.items() if v is not None}


WORKLOADS = {"fsmn_c3": fsmn_c3, "marblenet_c4": marblenet_c4, "firered_c5": firered_c5, "dfsmn_c5": dfsmn_c5}


def run_all(torch, device, reps=3, cpu_budget_s=3.0, only=None, log=lambda m: None):
    """-> {'fsmn_c3': {...}, 'marblenet_c4': {...}, 'marblenet_c4_one_gpu_share': {...}, 'firered_c5': {...},
    'dfsmn_c5': {...}}; a workload that fails is reported as {'error': ...} instead of taking the headline line down."""
    out = {}
    for name, fn in WORKLOADS.items():
        if only and name.split("_")[0] not in only:
            continue
        try:
            out[name] = fn(torch, device, reps, cpu_budget_s, log=log)
            if name == "marblenet_c4":
                share = marblenet_c4(torch, device, reps, 0, clips=1024, log=log, tag=None)      # the committed PMC passes are of the 8192-clip batch
                share["workload"] += " -- one GPU's share of the 8-GPU config"
                out["marblenet_c4_one_gpu_share"] = share
        except Exception as e:                               # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
        log(f"{name} done")
    if not only:
        # whole-config decision records: the default arithmetic against the float32-MFMA set on the FULL BASELINE workloads (decision_records.py)
        import decision_records
        out["decision_records"] = decision_records.run_all(torch, device, log=log)
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import torch
    import vadx  # noqa: F401
    t0 = time.perf_counter()
    out = run_all(torch, torch.device("cuda", 0), args.reps, 0 if args.no_cpu_baseline else 3.0,
                  [s for s in args.only.split(",") if s] or None,
                  log=lambda m: print(f"[bench_models +{time.perf_counter() - t0:6.1f}s] {m}", file=sys.stderr, flush=True))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
