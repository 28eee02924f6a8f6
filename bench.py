#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on BASELINE config[1]:
Silero-VAD f32, batch = 4096 synthetic 10 s @ 16 kHz clips per MI355X (weak scaling: every rank owns
its own 4096 clips, no data-path collective), raw audio resident in HBM -> speech-segment tables.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 without a launcher: this process starts N ranks itself (`python -m torch.distributed.run`, rendezvous on
127.0.0.1) BEFORE anything touches a GPU and relays rank 0's line; under an external launcher (WORLD_SIZE set, the way
the driver runs it) it is one of the ranks.  One "step" = one pass of the hot path over the resident batch: encoder
kernel (STFT conv + conv stack + W_ih, f32 MFMA) -> persistent LSTM kernel -> device segmenter.  Rank 0 prints ONE JSON
line.  Besides the contract keys it carries
  * `roofline` / `roofline_recurrent`: the two matrix-pipe kernels against the f32 MFMA peak, with SURVEY 8(d) byte accounting
    (`algorithmic_bytes_per_launch` = 2052 B / window, `traffic` from the committed PMC passes, `traffic_ratio`);
  * `hbm`: the north star's HBM fraction of the whole step;
  * `feed`: the same batch timed INCLUDING the int16 upload from pinned host memory, double-buffered against compute;
  * `configs`: BASELINE configs 3, 4, 5 measured in the same process (bench_models.py), a few numbers each (N = 1 only);
  * `c4_sharded`: BASELINE config 4 as the config states it -- 8192 MarbleNet clips strong-sharded over the N GPUs, resident and
    from pinned host memory (every N);
  * `cpu_baseline`: the oracle driven like the reference drives ORT, on this host's cores (N = 1, rank 0), beside the reference's
    own published README figure.
The line is kept under 6 KB (the driver records its tail); everything it summarises -- per-entry-point splits, every roofline
object in full, sample descriptions -- is written by rank 0 to `--detail` (default gpurun_out/bench_detail.json: scratch;
the copy to keep is committed under profiles/ deliberately).
`--dry-run` replaces the device work by a sleep and RCCL by gloo so the launch / barrier / collect path can be tested
on a CPU-only box; its line says so (`data`: "dry-run") and carries no measurement.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIPS_PER_GPU = 4096
SAMPLES = 160000
WINDOW = 512
STEPS_PER_CLIP = (SAMPLES + WINDOW - 1) // WINDOW            # 313

# Algorithmic flops per 512-sample window as the reference network computes it on non-padding
# samples (DESIGN.md "Silero path"): MACs x 2
MAC_STFT = 4 * 258 * 256
MAC_CONV1 = 4 * 128 * 129 * 3
MAC_CONV2 = 2 * 64 * 128 * 3
MAC_CONV3 = 64 * 64 * 2
MAC_CONV4 = 128 * 64
MAC_IH = 512 * 128
MAC_HH = 512 * 128
FLOP_ENCODE = 2 * (MAC_STFT + MAC_CONV1 + MAC_CONV2 + MAC_CONV3 + MAC_CONV4 + MAC_IH)
FLOP_RECUR = 2 * MAC_HH
# What the encoder kernel actually issues per 16-window tile: v_mfma_f32_16x16x4 counts per phase
# (folded STFT 1024, Winograd F(4,3) conv1 1536, conv2 640, conv3 128, conv4 128, W_ih 1024), 2048 flop each.  Lower than
# the dense count because the DFT's time and frequency symmetries quarter the STFT contraction and conv1 runs as six
# Winograd-plane GEMMs instead of ten tap GEMMs (DESIGN.md "Silero path").
MFMA_PER_TILE = 1024 + 1536 + 640 + 128 + 128 + 1024
FLOP_ENCODE_ISSUED = MFMA_PER_TILE * 2048 // 16
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = f32 vector rate
# The split-product kernels (csrc/split3.h: every constant-weight GEMM as SIX v_mfma_f32_16x16x32_bf16 per K = 32 step on exactly split
# float32 operands) are priced against the f32-EQUIVALENT peak of that pipe: the dense bf16 MFMA peak / 6 -- never against 157.3.
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# split-product encoder (csrc/silero_split.hip), per 16-window tile: the folded STFT stays on f32 MFMAs (1024), everything else is
# bf16 MFMAs in groups of six: direct conv1 1920, conv2 480, conv3 96, conv4 96, W_ih 768 = 3360, i.e. 560 K = 32 steps of 16 x 16 outputs
SPLIT_F32_MFMA_PER_TILE = 1024
SPLIT_BF16_MFMA_PER_TILE = 1920 + 480 + 96 + 96 + 768
FLOP_SPLIT_F32_PART = SPLIT_F32_MFMA_PER_TILE * 2048 // 16                  # per window, on the f32 pipe
FLOP_SPLIT_BF16_PART = SPLIT_BF16_MFMA_PER_TILE // 6 * 16384 // 16          # per window, f32-equivalent flops of the split products
# fp16 x 2 kernels (csrc/split2.h, csrc/silero_h2.hip): THREE v_mfma_f32_16x16x32_f16 per K = 32 step, the folded STFT included -- priced
# against the dense fp16 MFMA peak / 3.  Per 16-window tile: STFT 480 (bin 64 as a fifth bin tile), conv1 960, conv2 240, conv3 48,
# conv4 48, W_ih 384 = 2160 MFMAs = 720 K = 32 steps of 16 x 16 outputs.
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense fp16 MFMA peak (= bf16)
PEAK_H2_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
H2_MFMA_PER_TILE = 480 + 960 + 240 + 48 + 48 + 384
FLOP_H2 = H2_MFMA_PER_TILE // 3 * 16384 // 16                               # per window, f32-equivalent flops
# What the chip sustains of a pipe's nominal peak when that pipe is kept busy (it clocks to its power budget): shader clock under a
# back-to-back MFMA stream / the 2.4 GHz the nominal peaks assume.  f32 and bf16: profiles/r04_dvfs_probe.txt (2.14 - 2.20 and 1.99 - 2.01
# GHz); fp16: profiles/r05_f16x2_probe.txt (2.07 - 2.15 GHz under three-MFMA groups).  `frac_of_sustained` = achieved / (peak x this).
SUSTAINED_OF_NOMINAL = {"f32": 0.87, "split": 0.80, "h2": 0.86}
SUSTAINED_SOURCE = "profiles/r04_dvfs_probe.txt (f32, bf16), profiles/r05_f16x2_probe.txt (fp16)"
ENC_KERNELS = {"f32": "silero_encode_kernel", "split": "silero_encode_split_kernel", "h2": "silero_encode_h2_kernel"}
REC_KERNELS = {"f32": "silero_lstm_kernel", "split": "silero_lstm_split_kernel", "h2": "silero_lstm_h2_kernel"}
ARITH_TEXT = {"f32": "float32 MFMAs (v_mfma_f32_16x16x4_f32)",
              "split": "float32 results from bf16 x 3 exact-split products on the bf16 matrix pipe (csrc/split3.h)",
              "h2": "float32 results from fp16 x 2 split products on the fp16 matrix pipe: operands to one float32 ulp by two round-to-nearest "
                    "fp16 terms, three MFMAs per K = 32 step, range-checked (csrc/split2.h)"}


def encoder_roofline(mode, frames, enc_ms):
    """(achieved TFLOP/s, peak TFLOP/s, flop per frame, note) of one encoder launch.  f32 kernel: issued f32-MFMA flops against 157.3.
    Split kernel: f32 flops of the STFT part + f32-EQUIVALENT flops of the split products, against the peak of that mix = total flops /
    (f32 part / 157.3 + split part / (2500 / 6)) -- the time the two matrix pipes need at their own peaks, back to back."""
    if mode == "h2":
        return frames * FLOP_H2 / (enc_ms * 1e-3) / 1e12, PEAK_H2_TFLOPS, FLOP_H2, (
            f"{FLOP_H2} f32-equivalent flops of fp16 x 2 split products per frame (folded STFT included); peak = 2500 / 3 TFLOP/s")
    if mode != "split":
        return frames * FLOP_ENCODE_ISSUED / (enc_ms * 1e-3) / 1e12, PEAK_F32_MFMA_TFLOPS, FLOP_ENCODE_ISSUED, "issued f32-MFMA flops"
    fl = FLOP_SPLIT_F32_PART + FLOP_SPLIT_BF16_PART
    t_min = FLOP_SPLIT_F32_PART / PEAK_F32_MFMA_TFLOPS + FLOP_SPLIT_BF16_PART / PEAK_SPLIT_TFLOPS          # 1e-12 s per frame
    return frames * fl / (enc_ms * 1e-3) / 1e12, fl / t_min, fl, (
        f"{FLOP_SPLIT_F32_PART} f32-MFMA flops (STFT) + {FLOP_SPLIT_BF16_PART} f32-equivalent flops of bf16 x 3 split products per frame; peak = "
        "that mix at 157.3 and 2500 / 6 TFLOP/s")
PEAK_HBM_GBPS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec peak
# algorithmic HBM bytes per 512-sample window on the Silero path = 2048 B of float32 PCM in (as the reference
# feeds it) + one 4-byte score out.  The encoder -> LSTM intermediate `gx` (2048 B written + 2048 B read per window) is
# DESIGN traffic, not algorithmic: it is what `traffic_ratio` exposes.
ALGO_BYTES_PER_WINDOW = 2048 + 4
FEED_CHUNK_CLIPS = 512                # feed-inclusive mode: upload granularity (164 MB of int16 per chunk)


def _kernel_counter(path, kernel, counter):
    """mean per-dispatch value of one PMC counter for the LARGEST grid of `kernel` in a profiles/*/SUMMARY.txt"""
    best, grid = None, -1
    for line in open(path):
        if kernel not in line or "grid=" not in line or f" {counter} " not in line:
            continue
        g = int(line.split("grid=")[1].split()[0])
        if g >= grid:
            grid, best = g, float(line.split()[-2])
    return best, grid


def profiled_traffic(kernel="silero_encode_kernel"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/rNN*/SUMMARY.txt, written by
    tools/profile_bench.sh from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this very command).  FETCH_SIZE is
    doubled (gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md "HBM"); both are KiB."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "SUMMARY.txt"))):
        fetch, g1 = _kernel_counter(path, kernel, "FETCH_SIZE")
        write, g2 = _kernel_counter(path, kernel, "WRITE_SIZE")
        if fetch is not None and write is not None and g1 == g2:
            best = {"bytes": (2.0 * fetch + write) * 1024.0, "source": os.path.relpath(path, ROOT), "grid_threads": g1}
    return best


def profiled_launch_ms(kernel="silero_encode_kernel"):
    """n / mean / median launch duration [ms] of the LARGEST grid of `kernel` in the newest committed profile (the "per (kernel, grid)
    dispatch durations" block tools/profile_bench.sh writes from the kernel trace).  Reported BESIDE the line's own figure: the line's
    `roofline.achieved` uses the mean of THIS run's HIP events (`launch_ms.basis`), the profile is another run on another box."""
    import glob
    import re
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "SUMMARY.txt"))):
        grid = -1
        for line in open(path):
            m = re.search(r"grid=\s*(\d+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)\s+median=\s*(\d+)", line)
            if m and kernel in line and int(m.group(1)) >= grid:
                grid = int(m.group(1))
                best = {"n": int(m.group(2)), "mean": float(m.group(3)) * 1e-6, "median": float(m.group(4)) * 1e-6,
                        "source": os.path.relpath(path, ROOT)}
    return best


# Per-SIMD instruction costs beside f32 MFMAs, measured with two or more waves per SIMD (tools/mfma_valu_overlap.sh):
# on gfx950 the f32-input MFMA shares the vector datapath, so VALU time ADDS to MFMA time instead of hiding under it.
NS_PER_MFMA, NS_PER_VALU = 13.7, 0.93


def instruction_mix(kernel="silero_encode_kernel", arithmetic="f32"):
    """MFMA / VALU instruction counts per launch of `kernel` (SQ_INSTS_* of the newest committed PMC summary) and -- for f32-MFMA kernels
    ONLY -- the fraction of the f32-MFMA peak that mix could reach if nothing but issue time were spent: MFMA / (MFMA + VALU * 0.93 / 13.7).
    The additive model was measured for f32-input MFMAs, which share the vector datapath; beside bf16 / fp16 MFMAs VALU work hides
    (tools/bf16x3_probe.sh, tools/f16x2_probe.sh), so a split-product kernel carries the counts and NO model ceiling."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "SUMMARY.txt"))):
        mfma, g1 = _kernel_counter(path, kernel, "SQ_INSTS_MFMA")
        valu, g2 = _kernel_counter(path, kernel, "SQ_INSTS_VALU")
        if mfma and valu and g1 == g2:
            best = {"mfma_insts": mfma, "valu_insts": valu, "valu_per_mfma": valu / mfma, "arithmetic": arithmetic,
                    "ceiling_frac": mfma * NS_PER_MFMA / (mfma * NS_PER_MFMA + valu * NS_PER_VALU),
                    "model": f"{NS_PER_MFMA} ns per v_mfma_f32_16x16x4_f32 + {NS_PER_VALU} ns per VALU instruction per SIMD, additive "
                             "(tools/mfma_valu_overlap.sh)", "source": os.path.relpath(path, ROOT)}
            if arithmetic != "f32":
                best["ceiling_frac"] = None
                best["model"] = "none: VALU instructions hide beside bf16 / fp16 MFMAs (tools/bf16x3_probe.sh, tools/f16x2_probe.sh)"
    return best


def enc_kernel_name():
    from vadx import silero
    return ENC_KERNELS[silero.encoder_mode()]


def synth_batch(torch, device, batch, samples, seed, pcm16=False):
    """int16-quantised burst clips generated on the GPU (every clip unique): 0.5-2 s segments
    alternating N(0,3000) / N(0,30), then x 1/32768 as the reference feeds Silero
    (Silero/Inference_Silero_VAD_ONNX.py:83).  pcm16=True returns the int16 samples themselves."""
    import bench_models
    pcm = bench_models.synth_pcm16(torch, device, batch, samples, seed)
    if pcm16:
        return pcm
    out = torch.empty((batch, samples), dtype=torch.float32, device=device)
    for b0 in range(0, batch, 512):
        out[b0:b0 + 512] = pcm[b0:b0 + 512].to(torch.float32) * 0.000030517578
    return out


def cpu_baseline(budget_s=10.0):
    """The oracle (torch-CPU restatement of the reference graph) driven as the reference drives ORT:
    batch 1, one call per 512-sample window, state carried -- timed on this host's cores.
    Batch-1 windows are tiny ops, so more threads is not faster: a short calibration picks the
    best intra-op thread count (the reference uses ORT's auto setting / physical cores)."""
    import torch
    from oracle import silero as osil
    import bench_models
    import vadx  # noqa: F401
    from vadx import weights
    w = {k: torch.from_numpy(v) for k, v in weights.silero_synthetic(1234).items()}
    clips = weights.burst_clips(4, SAMPLES, seed=4321).astype(np.float32) * np.float32(0.000030517578)
    model = osil.OnnxWrapperOracle(w)

    def run(seconds):
        """windows/s over ~`seconds`, checked every window so a slow host cannot overrun"""
        model.reset_states()
        n, t0 = 0, time.perf_counter()
        with torch.no_grad():
            while True:
                a = clips[(n // STEPS_PER_CLIP) % 4]
                s = (n % STEPS_PER_CLIP) * WINDOW
                chunk = torch.from_numpy(a[s:s + WINDOW])
                if chunk.shape[0] < WINDOW:
                    chunk = torch.nn.functional.pad(chunk, (0, WINDOW - chunk.shape[0]))
                if n % STEPS_PER_CLIP == 0:
                    model.reset_states()
                model(chunk, 16000).item()
                n += 1
                el = time.perf_counter() - t0
                if el >= seconds:
                    return n / el, n, el

    ncpu = os.cpu_count() or 1
    best_thr, best_rate = 1, 0.0
    for thr in sorted({1, 2, 4, 8, min(16, ncpu)}):
        if thr > ncpu:
            continue
        torch.set_num_threads(thr)
        run(0.2)
        rate, _, _ = run(0.8)
        if rate > best_rate:
            best_thr, best_rate = thr, rate
    torch.set_num_threads(best_thr)
    rate, n, el = run(budget_s)
    # BASELINE.md section 3 protocol beside it: one 10 s clip, 3 warm-up + 10 timed repetitions, median; model calls only (the
    # reference's convention) and end to end with the segmenter.  Threads: the calibrated count above, NOT every host thread --
    # batch-1 windows are tiny ops and torch's intra-op pool on a 256-thread host spends milliseconds per op synchronising
    # (that variant ran for over half an hour on the GPU box).  A wall-clock guard keeps a slow host from overrunning.
    from oracle import postproc as opp
    clip = torch.from_numpy(clips[0])
    guard_t0, guard_s = time.perf_counter(), 30.0

    def one_clip(with_post):
        t0 = time.perf_counter()
        with torch.no_grad():
            pr = osil.speech_probs(clip, model, 16000)
        t1 = time.perf_counter()
        if with_post:
            opp.silero_segments([float(v) for v in pr], SAMPLES, threshold=0.5, max_speech_duration_s=20, min_speech_duration_ms=250,
                                min_silence_duration_ms=250, return_seconds=True)
            t1 = time.perf_counter()
        return t1 - t0

    def timed(with_post, reps):
        out = []
        for _ in range(reps):
            if out and time.perf_counter() - guard_t0 > guard_s:
                break
            out.append(one_clip(with_post))
        return out

    warm = timed(False, 3)
    t_model, t_e2e = timed(False, 10), timed(True, 10)
    med_model, med_e2e = float(np.median(t_model)), float(np.median(t_e2e))
    protocol = {"threads": best_thr, "repetitions": f"{len(warm)} warm-up + {len(t_model)} / {len(t_e2e)} timed, median",
                "clip": "one synthetic 10 s clip, batch 1",
                "model_calls_only": {"value": STEPS_PER_CLIP / med_model, "rtf": med_model / (SAMPLES / 16000.0)},
                "end_to_end": {"value": STEPS_PER_CLIP / med_e2e, "rtf": med_e2e / (SAMPLES / 16000.0)}}
    # the same oracle batched over clips on all cores (NOT how the reference runs; shown for scale)
    torch.set_num_threads(min(ncpu, 64))
    bb = 64
    xb = torch.from_numpy(weights.burst_clips(bb, 16 * WINDOW, seed=99).astype(np.float32) * np.float32(0.000030517578))
    m2 = osil.OnnxWrapperOracle(w)
    with torch.no_grad():
        m2.audio_forward(xb[:, :2 * WINDOW], 16000)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 4.0:
            m2.audio_forward(xb, 16000)
            reps += 1
        batched = reps * bb * 16 / (time.perf_counter() - t0)
    return {"value": rate, "unit": "frames/s", "cores": best_thr, "kind": "port", "cpu": bench_models.cpu_model(),
            "sample": f"{n} windows of synthetic 10 s clips, batch 1, one call per 512-sample window, state carried "
                      f"(torch-CPU oracle stand-in for ORT-CPU; best of 1/2/4/8/16 intra-op threads on a "
                      f"{ncpu}-CPU host), {el:.1f} s",
            "baseline_md_protocol": protocol,
            "batched_value": batched, "batched_note": f"same oracle, batch {bb}, {min(ncpu, 64)} threads (not the reference's mode)"}


def compact_line(full, detail_path=None):
    """The driver's record keeps the TAIL of the line, so the line stays short (< 6 KB): the contract keys, the dominant kernel's
    roofline with its byte accounting, one small object per BASELINE config, the CPU baseline.  `full` (every number) is what
    `--detail` holds."""
    import bench_models
    r4 = lambda v: None if v is None else float(f"{v:.5g}")      # noqa: E731
    ro, rr, hb, fd, cp = full["roofline"], full["roofline_recurrent"], full["hbm"], full.get("feed"), full.get("cpu_baseline")
    mix = ro.get("instruction_mix") or {}
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config", "per_gpu_value", "rtf_batch1")}
    line["kernel_ms"] = {k: r4(v) for k, v in full["kernel_ms"].items()}
    if full.get("ms_per_step_by_arithmetic"):
        line["ms_per_step_by_arithmetic"] = {k: r4(v) for k, v in full["ms_per_step_by_arithmetic"].items()}
    line["roofline"] = {"bound": ro["bound"], "kernel": ro["kernel"], "achieved": ro["achieved"], "peak": ro["peak"], "unit": ro["unit"],
                        "frac": ro["frac"], "traffic": ro["traffic"], "traffic_unit": ro["traffic_unit"],
                        "algorithmic_bytes_per_launch": ro["algorithmic_bytes_per_launch"], "traffic_ratio": r4(ro["traffic_ratio"]),
                        "flop_per_frame": ro["flop_per_frame"], "dense_equivalent_achieved": r4(ro["dense_equivalent"]["achieved"]),
                        "valu_per_mfma": r4(mix.get("valu_per_mfma")), "ceiling_frac": r4(mix.get("ceiling_frac")),
                        "traffic_source": ro.get("traffic_source")}
    if full.get("encoder_launch_ms"):
        lm = full["encoder_launch_ms"]
        line["roofline"]["launch_ms"] = {k: (r4(v) if isinstance(v, float) else v) for k, v in lm.items() if k not in ("basis", "committed_profile")}
        line["roofline"]["launch_ms"]["basis"] = "mean of this run's HIP events"
        cp_ = lm.get("committed_profile")
        if cp_:
            line["roofline"]["launch_ms"]["profile"] = {"n": cp_["n"], "mean": r4(cp_["mean"]), "median": r4(cp_["median"])}
    line["roofline_recurrent"] = {"kernel": rr["kernel"], "achieved": r4(rr["achieved"]), "frac": r4(rr["frac"])}
    line["hbm"] = {"algorithmic_bytes_per_frame": hb["algorithmic_bytes_per_frame"], "achieved_GBps": r4(hb["achieved_GBps"]),
                   "peak_GBps": hb["peak_GBps"], "frac": r4(hb["frac"]), "traffic_ratio": r4(hb["traffic_ratio"])}
    if fd is not None:
        line["feed"] = fd if "error" in fd else {"value": r4(fd["value"]), "ms_per_step": r4(fd["ms_per_step"]),
                                                 "upload_alone_GBps": r4(fd["upload_alone_GBps"]), "pcie_peak_GBps": fd["pcie_peak_GBps"],
                                                 "scores_bit_identical_to_resident_f32_path": fd["scores_bit_identical_to_resident_f32_path"]}
    if full.get("secondary"):
        line["configs"] = {k: bench_models.compact(v) for k, v in full["secondary"].items()}
    c4 = full.get("c4_sharded")
    if c4 is not None:
        line["c4_sharded"] = c4 if "error" in c4 else {
            "clips": c4["clips"], "n_gpus": c4["n_gpus"], "scaling": "strong", "ms": r4(c4["ms"]), "frames_per_s": r4(c4["frames_per_s"]),
            "feed_ms": r4((c4.get("feed") or {}).get("ms")), "feed_frames_per_s": r4((c4.get("feed") or {}).get("frames_per_s")),
            "feed_bit_identical": (c4.get("feed") or {}).get("scores_bit_identical_to_resident")}
    if cp is not None:
        pub = cp.get("reference_published") or {}
        line["cpu_baseline"] = {"value": r4(cp["value"]), "unit": cp["unit"], "cores": cp["cores"], "kind": cp["kind"], "cpu": cp["cpu"],
                                "sample": f"{cp['sample'].split(',')[0]}, batch 1, one oracle call per 512-sample window (torch-CPU stand-in for ORT-CPU)",
                                "reference_published": {"frames_per_s": r4(pub.get("frames_per_s")), "rtf": pub.get("rtf"),
                                                        "hardware": pub.get("hardware"), "source": pub.get("source")}}
    else:
        line["cpu_baseline"] = None
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path
    return line


_T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU (default = BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 3-5 (bench_models.py)")
    ap.add_argument("--no-feed", action="store_true", help="skip the PCIe-inclusive (pinned, double-buffered upload) measurement")
    ap.add_argument("--secondary-reps", type=int, default=3)
    ap.add_argument("--no-c4-sharded", action="store_true", help="skip BASELINE config 4 strong-sharded over the ranks")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where rank 0 writes the full (long) result object; '' = nowhere")
    ap.add_argument("--dry-run", action="store_true", help="CPU-only: gloo + sleep instead of RCCL + kernels (launch-path test)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`bench.py --gpus N` started plainly: become the launcher.  No torch import, no HIP call has happened in this
    process -- the ranks are fresh children of torch.distributed.run -- and this process only relays their output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("self-launch: " + " ".join(cmd))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """The N-rank launch / barrier / max-over-ranks / rank-0-prints path without a GPU (gloo): tests/test_bench_contract.py."""
    import vadx  # noqa: F401
    from vadx import shard
    rank, _local, world = shard.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = shard.init("gloo")
    for _ in range(args.warmup):
        time.sleep(0.002)
    shard.fence(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.005)
    shard.fence(dist)
    elapsed = shard.max_over_ranks(dist, time.perf_counter() - t0)
    lo, hi = shard.shard_bounds(world * args.clips, rank, world)
    owned = shard.gather_ragged(dist, [(rank, lo, hi)])
    c4 = shard.gather_ragged(dist, [(rank,) + tuple(shard.shard_bounds(8192, rank, world))])      # BASELINE config 4: strong shards
    if rank == 0:
        print(json.dumps({"metric": "audio frames/sec/GPU (16 kHz, 512-sample hop); RTF at batch=1", "value": None,
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "dry-run (no GPU work: launch / barrier / collect path only)",
                          "config": {"workload": "dry-run", "clips_per_gpu": args.clips, "shards": owned},
                          "c4_sharded": {"clips": 8192, "scaling": "strong", "shards": c4}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


class FeedPipeline:
    """The headline batch timed from PINNED HOST int16 PCM through the product's own feed (vadx.silero.HostFeed: chunks of
    FEED_CHUNK_CLIPS clips cross PCIe on a copy stream into one of two device buffers while the encoder of the previous chunk
    runs); the recurrent kernel and the segmenter run once over the whole batch."""

    def __init__(self, torch, eng, L, host_pcm, probs, segs, counts, lens, prm, cap):
        self.t, self.eng, self.L = torch, eng, L
        self.host = host_pcm
        B, N = host_pcm.shape
        self.feed = eng.host_feed(B, N, FEED_CHUNK_CLIPS)
        self.B, self.N, self.T = B, N, self.feed.T
        self.out = (probs, segs, counts, lens, prm, cap)

    def step(self):
        t, L, eng = self.t, self.L, self.eng
        probs, segs, counts, lens, prm, cap = self.out
        st = C.c_void_p(t.cuda.current_stream().cuda_stream)
        from vadx import _lib
        self.feed.encode(self.host)
        eng.recur(self.B, self.T, probs)
        _lib.check(L.vadx_silero_segments(probs.data_ptr(), self.B, self.T, lens.data_ptr(), C.byref(prm), segs.data_ptr(),
                                          counts.data_ptr(), cap, st))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    if args.dry_run:
        return dry_run(args)

    import torch
    import bench_models
    import vadx  # noqa: F401
    from vadx import _lib, shard, silero, weights

    rank, local_rank, world = shard.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or run `bench.py --gpus N` "
                         "without a launcher and let it start the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the vadx product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = shard.init("nccl", device)        # None for a single process; RCCL otherwise (barrier/timing only)

    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(device)}; torch imported")
    B = args.clips
    eng = silero.SileroEngine(weights.silero_synthetic(1234), device=device)
    pcm = synth_batch(torch, device, B, SAMPLES, seed=1234 + rank, pcm16=True)
    audio = torch.empty((B, SAMPLES), dtype=torch.float32, device=device)          # resident in HBM, as the reference feeds it
    for b0 in range(0, B, 512):
        audio[b0:b0 + 512] = pcm[b0:b0 + 512].to(torch.float32) * 0.000030517578
    torch.cuda.synchronize()
    log(f"{B} x {SAMPLES} synthetic clips resident ({audio.numel() * 4 / 2**30:.2f} GiB)")
    L = _lib.lib()
    T = STEPS_PER_CLIP
    probs = torch.empty((B, T), dtype=torch.float32, device=device)
    cap = 64
    segs = torch.empty((B, cap, 2), dtype=torch.int64, device=device)
    counts = torch.empty((B,), dtype=torch.int32, device=device)
    lens = torch.full((B,), SAMPLES, dtype=torch.int64, device=device)
    prm = silero.seg_params(threshold=0.5, max_speech_duration_s=20, min_speech_duration_ms=250,
                            min_silence_duration_ms=250)     # Silero/Inference_Silero_VAD_ONNX.py:88-96
    ws = eng._workspace(B, T)
    st = _lib.stream_ptr()
    cfg = eng.cfg()                     # the arithmetic rides with every call (include/vadx.h: vadx_silero_cfg)

    flag_w, amax_w = C.c_uint32(0), C.c_float(0.0)
    guard = eng.mode()
    range_flags_seen = [0]

    def step(ev=None, cfg=cfg, guarded=(guard == "h2")):
        if ev:
            ev[0].record()
        _lib.check(L.vadx_silero_encode(eng.packed.data_ptr(), audio.data_ptr(), B, SAMPLES, _lib.row_stride(audio),
                                        ws.data_ptr(), ws.numel(), st, cfg))
        if ev:
            ev[1].record()
        _lib.check(L.vadx_silero_recur(eng.packed.data_ptr(), ws.data_ptr(), ws.numel(), B, T, None,
                                       probs.data_ptr(), None, st, cfg))
        if ev and len(ev) > 2:
            ev[2].record()
        _lib.check(L.vadx_silero_segments(probs.data_ptr(), B, T, lens.data_ptr(), C.byref(prm), segs.data_ptr(),
                                          counts.data_ptr(), cap, st))
        if ev and len(ev) > 2:
            ev[3].record()
        if guarded:
            # the fp16 x 2 arithmetic's range protocol is part of the step, as in SileroEngine._guarded and tests/c/cabi_silero.c: 8 bytes back
            # + one stream synchronisation per step (the timed step used to skip it)
            _lib.check(L.vadx_silero_range_flag(eng.packed.data_ptr(), 1, C.byref(flag_w), C.byref(amax_w), st))
            range_flags_seen[0] += int(flag_w.value != 0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("warm-up done")

    def fence():
        shard.fence(dist, torch.cuda.synchronize)

    # the timed steps carry the HIP events of the DOMINANT kernel only (roofline.achieved is its mean launch duration over the timed region);
    # the other two kernels' durations come from an event-bracketed pass right behind it: every event between two launches is a marker the queue
    # waits on, and with the per-step flag read draining the queue their cost (0.1 - 0.2 ms per step) would sit in the headline
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(dist, elapsed, device)

    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    enc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    ev4 = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(max(3, min(args.steps, 10)))]
    for e4 in ev4:
        step(e4, cfg, False)
    torch.cuda.synchronize()
    rec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev4]))
    seg_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev4]))
    # (the contract's `roofline.achieved` is on the AVERAGE launch duration; the median and the extremes go to the detail file so that a
    #  profile taken on another box can be compared with the middle of this run, not with a mean that one slow launch moved)
    enc_all = sorted(e[0].elapsed_time(e[1]) for e in events)
    enc_stats = {"mean": enc_ms, "median": float(np.median(enc_all)), "min": enc_all[0], "max": enc_all[-1], "launches": len(enc_all),
                 "basis": "roofline.achieved uses `mean` = the average of this run's HIP-event launch durations",
                 "committed_profile": profiled_launch_ms(enc_kernel_name()) if (B, T) == (CLIPS_PER_GPU, STEPS_PER_CLIP) else None}
    assert int(counts.max().item()) <= cap, "segment table overflow"
    assert bool(torch.isfinite(probs).all())
    range_flag = eng.range_flag()       # fp16 x 2 kernels: no activation of the timed steps may have left the fp16 range
    assert range_flag[0] == 0 and range_flags_seen[0] == 0, f"fp16 range flag raised during the timed steps: {range_flag}, {range_flags_seen}"
    n_seg = int(counts.sum().item())
    probs_resident = probs.clone()
    # the same step on every arithmetic, same run, same buffers: float32 MFMAs, bf16 x 3 (operands exact to float32's 24
    # bits, float32's exponent range -- what a flagged batch is recomputed on) and fp16 x 2 with and without its per-step flag read
    by_arith = {}
    if not args.dry_run:
        for name, mode, guarded in (("f32", "f32", False), ("bf16x3", "split", False), ("f16x2", "h2", True), ("f16x2_without_flag_read", "h2", False)):
            if mode == "h2" and not eng.h2_ok:
                continue
            c_ = eng.cfg(mode)
            for _ in range(2):
                step(None, c_, guarded)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(args.steps):
                step(None, c_, guarded)
            torch.cuda.synchronize()
            by_arith[name] = (time.perf_counter() - ta) / args.steps * 1e3
        step(None, cfg, False)          # leave probs / segments as the headline arithmetic computed them
        torch.cuda.synchronize()
        assert bool(torch.equal(probs, probs_resident))
        log("ms/step by arithmetic: " + ", ".join(f"{k} {v:.3f}" for k, v in by_arith.items()))

    frames_per_step = world * B * T
    value = frames_per_step * args.steps / elapsed
    enc_mode = silero.encoder_mode()
    enc_kernel, rec_kernel = ENC_KERNELS[enc_mode], REC_KERNELS[enc_mode]
    achieved, enc_peak, enc_flop, enc_note = encoder_roofline(enc_mode, B * T, enc_ms)
    rec_peak = {"f32": PEAK_F32_MFMA_TFLOPS, "split": PEAK_SPLIT_TFLOPS, "h2": PEAK_H2_TFLOPS}[enc_mode]
    sustained = SUSTAINED_OF_NOMINAL[enc_mode]

    # ---- the same batch from pinned host int16, upload overlapped with compute (every rank feeds its own GPU at once)
    feed = None
    if not args.no_feed:
        # rank-local set-up (a pinned allocation can be refused) first, then the ranks AGREE before the section with barriers and
        # all-reduces: a rank that bailed out alone would leave the others waiting in them (shard.all_ok)
        err, pipe, host = None, None, None
        try:
            host = torch.empty((B, SAMPLES), dtype=torch.int16, pin_memory=True)
            host.copy_(pcm)
            torch.cuda.synchronize()
            del audio
            torch.cuda.empty_cache()
            pipe = FeedPipeline(torch, eng, L, host, probs, segs, counts, lens, prm, cap)
            for _ in range(max(1, args.warmup)):
                pipe.step()
            torch.cuda.synchronize()
        except Exception as e:                                       # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if shard.all_ok(dist, err is None, device):
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pipe.step()
            fence()
            feed_el = shard.max_over_ranks(dist, time.perf_counter() - t1, device)
            same = bool(torch.equal(probs, probs_resident))          # int16 in-kernel scaling == host-side float32 product, bit for bit
            up_bytes = B * SAMPLES * 2
            # upload alone (same pinned buffer, same chunking, nothing overlapped): what PCIe gives this process
            t2 = time.perf_counter()
            for b0 in range(0, B, FEED_CHUNK_CLIPS):
                pipe.feed.buf[0][:min(FEED_CHUNK_CLIPS, B - b0)].copy_(host[b0:b0 + FEED_CHUNK_CLIPS], non_blocking=True)
            torch.cuda.synchronize()
            up_s = time.perf_counter() - t2
            feed = {"value": world * B * T * args.steps / feed_el, "unit": "frames/s", "ms_per_step": feed_el / args.steps * 1e3,
                    "upload": "int16 PCM from pinned host memory, chunks of %d clips double-buffered against the encoder" % FEED_CHUNK_CLIPS,
                    "upload_bytes_per_step_per_gpu": up_bytes, "upload_alone_ms": up_s * 1e3,
                    "upload_alone_GBps": up_bytes / up_s / 1e9, "pcie_peak_GBps": 63.0,
                    "scores_bit_identical_to_resident_f32_path": same}
            log(f"feed-inclusive: {feed['ms_per_step']:.2f} ms/step, upload alone {up_s * 1e3:.1f} ms ({feed['upload_alone_GBps']:.1f} GB/s)")
        else:
            feed = {"error": err or "another rank could not set its host feed up"}
            log(f"feed-inclusive mode skipped on every rank: {feed['error']}")
        del pipe, host
    del pcm

    rtf_b1 = None
    cpu = None
    secondary = None
    if rank == 0:
        one = synth_batch(torch, device, 1, SAMPLES, seed=99)
        eng.clips(one)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.clips(one)
        torch.cuda.synchronize()
        rtf_b1 = (time.perf_counter() - t1) / 5 / (SAMPLES / 16000.0)
        log(f"rtf_batch1 = {rtf_b1:.5f}")
    torch.cuda.empty_cache()
    eng._ws = None
    c4s = None
    if not args.no_c4_sharded:                               # every rank: its shard of the 8192 clips
        try:
            c4s = bench_models.marblenet_c4_sharded(torch, device, dist, rank, world, max(2, args.secondary_reps), log=log,
                                                    feed=not args.no_feed)
        except Exception as e:                               # noqa: BLE001
            if dist is not None:                             # the leg holds barriers: a rank that swallowed its error would hang the rest
                raise
            c4s = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if world == 1 and not args.no_secondary:
        secondary = bench_models.run_all(torch, device, args.secondary_reps, 0 if args.no_cpu_baseline else 3.0, log=log)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        log("cpu baseline done")

    if rank == 0:
        full = (B, T) == (CLIPS_PER_GPU, STEPS_PER_CLIP)
        tr_enc = profiled_traffic(enc_kernel) if full else None
        tr_rec = profiled_traffic(rec_kernel) if full else None
        # the additive VALU : MFMA cost model was measured for f32-input MFMAs (they share the vector datapath); beside bf16 MFMAs VALU work
        # hides (tools/bf16x3_probe.sh), so the split kernels carry the counts without a model ceiling
        mix_enc = instruction_mix(enc_kernel, enc_mode) if full else None
        algo_launch = B * T * ALGO_BYTES_PER_WINDOW
        step_traffic = (tr_enc["bytes"] + tr_rec["bytes"]) if (tr_enc and tr_rec) else None
        step_s = elapsed / args.steps
        line = {
            "metric": "audio frames/sec/GPU (16 kHz, 512-sample hop); RTF at batch=1",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Silero-VAD f32, batch=4096 synthetic 10 s @16 kHz clips per GPU "
                                   "(STFT conv + conv1d stack + LSTM cell HIP, seeded synthetic weights)",
                       "arithmetic": ARITH_TEXT[enc_mode],
                       "clips_per_gpu": B, "samples_per_clip": SAMPLES, "frames_per_clip": T,
                       "parallelism": f"clip-sharded x{world}, no collective"},
            "per_gpu_value": value / world, "rtf_batch1": rtf_b1,
            "kernel_ms": {enc_kernel: enc_ms, rec_kernel: rec_ms, "silero_segments_kernel": seg_ms},
            "ms_per_step_by_arithmetic": by_arith,
            "range_protocol": ("the timed step reads vadx_silero_range_flag after its launches (8 B + a stream synchronisation), as every caller of "
                               "the fp16 x 2 arithmetic does" if enc_mode == "h2" else "none (this arithmetic has float32's exponent range)"),
            "encoder_launch_ms": enc_stats,
            "segments_found": n_seg,
            # achieved = the flops the kernel's algorithm needs (MFMA-issued: folded DFT, no padding taps) / its time;
            # the reference's dense arithmetic would count FLOP_ENCODE per frame ("dense_equivalent").
            # Bytes follow SURVEY 8(d): algorithmic = 2048 B f32 PCM in + 4 B score out per window; `traffic` = the encoder
            # launch's counted HBM bytes (it also writes the 2048 B/window gx intermediate the LSTM kernel re-reads).
            "roofline": {"bound": "mfma", "kernel": enc_kernel, "achieved": achieved,
                         "peak": enc_peak, "unit": "TFLOP/s", "frac": achieved / enc_peak,
                         "frac_of_sustained": achieved / (enc_peak * sustained), "sustained_of_nominal": sustained,
                         "sustained_source": SUSTAINED_SOURCE,
                         "arithmetic": enc_mode, "flops_counted": enc_note,
                         "traffic": tr_enc["bytes"] if tr_enc else None, "traffic_unit": "B/launch",
                         "traffic_source": tr_enc["source"] if tr_enc else None,
                         "algorithmic_bytes_per_launch": algo_launch,
                         "traffic_ratio": (tr_enc["bytes"] / algo_launch) if tr_enc else None,
                         "flop_per_frame": enc_flop, "frames_per_launch": B * T,
                         "dense_equivalent": {"flop_per_frame": FLOP_ENCODE,
                                              "achieved": achieved * FLOP_ENCODE / enc_flop},
                         # what this kernel's own instruction mix allows of the peak (VALU time adds to f32-MFMA time on gfx950)
                         "instruction_mix": mix_enc},
            # the recurrent kernel is matrix-pipe work too (W_hh x h, 16 clips = one MFMA tile wide): its own fraction
            "roofline_recurrent": {"bound": "mfma", "kernel": rec_kernel, "flop_per_frame": FLOP_RECUR,
                                   "achieved": B * T * FLOP_RECUR / (rec_ms * 1e-3) / 1e12, "peak": rec_peak, "arithmetic": enc_mode,
                                   "unit": "TFLOP/s", "frac": B * T * FLOP_RECUR / (rec_ms * 1e-3) / 1e12 / rec_peak,
                                   "frac_of_sustained": B * T * FLOP_RECUR / (rec_ms * 1e-3) / 1e12 / (rec_peak * sustained),
                                   "traffic": tr_rec["bytes"] if tr_rec else None,
                                   "instruction_mix": instruction_mix(rec_kernel, enc_mode) if full else None},
            # the north star's HBM view of the whole step: SURVEY 8(d) algorithmic bytes / step time against 8 TB/s, and what
            # the step really moves (encoder + LSTM launches, PMC) over the algorithmic bytes
            "hbm": {"algorithmic_bytes_per_frame": ALGO_BYTES_PER_WINDOW, "algorithmic_bytes_per_step": algo_launch,
                    "achieved_GBps": algo_launch / step_s / 1e9, "peak_GBps": PEAK_HBM_GBPS,
                    "frac": algo_launch / step_s / 1e9 / PEAK_HBM_GBPS,
                    "traffic_bytes_per_step": step_traffic,
                    "traffic_ratio": (step_traffic / algo_launch) if step_traffic else None,
                    "traffic_GBps": (step_traffic / step_s / 1e9) if step_traffic else None},
            "feed": feed,
            "secondary": secondary,
            "c4_sharded": c4s,
            "cpu_baseline": cpu,
        }
        if cpu is not None:
            cpu["reference_published"] = bench_models.REFERENCE_PUBLISHED["silero"]
        if args.detail:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
                with open(args.detail, "w") as fh:
                    json.dump(line, fh, indent=1)
            except OSError as e:
                log(f"detail file not written: {e}")
        print(json.dumps(compact_line(line, args.detail)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
