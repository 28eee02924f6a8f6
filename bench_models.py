#!/usr/bin/env python
"""Secondary measurements (NOT the driver's bench contract -- that is bench.py): device time of the
other BASELINE configs on one MI355X, inputs resident in HBM, seeded synthetic weights.
    python bench_models.py [--reps 5] > profiles/rNN_models.json"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def timed(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import vadx  # noqa: F401
    from vadx import firered, fsmn, marblenet, weights
    from vadx import timestamps as ts
    out = {}
    # ---- config 3: FSMN, 4096 x 10 s
    eng = fsmn.FsmnEngine(weights.fsmn_synthetic(1234))
    lb, stride = eng.grid()
    base = weights.burst_clips(64, 160000, seed=123)
    noise = np.random.default_rng(1).standard_normal((64, 20000))
    rows = np.stack([fsmn.pad_to_window_grid(ts.normalize_to_int16(base[b].astype(np.float32)), 16000, stride, noise[b]) for b in range(64)])
    W = (rows.shape[1] - 16000) // stride + 1
    big = torch.from_numpy(rows).cuda().repeat(64, 1)
    ms_feat = timed(torch, lambda: eng.features(big, W, stride), args.reps)
    ms_all = timed(torch, lambda: eng.flags(big, W), args.reps)
    out["fsmn_config3"] = {"clips": 4096, "seconds_per_clip": 10, "windows_per_clip": W, "ms_frontend_energy": ms_feat,
                           "ms_total": ms_all, "hop512_frames_per_s": 4096 * 313 / (ms_all * 1e-3),
                           "net_frames_10ms": 4096 * W * 101, "net_TFLOPs": 4096 * W * 101 * 0.854e6 / ((ms_all - ms_feat) * 1e-3) / 1e12}
    del big
    # ---- config 4 (one GPU's view): MarbleNet, 8192 x 89,431 samples
    mb = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234))
    base = weights.burst_clips(64, 89431, seed=55)
    big = torch.from_numpy(base).cuda().repeat(128, 1)
    fe = mb.frontend(89431)
    ms_fe = timed(torch, lambda: fe.logmel(big, 1, 89431), args.reps)
    ms_all = timed(torch, lambda: mb.run(big), args.reps)
    out["marblenet_config4_1gpu"] = {"clips": 8192, "samples_per_clip": 89431, "ms_frontend": ms_fe, "ms_total": ms_all,
                                     "hop512_frames_per_s": 8192 * 89431 / 512 / (ms_all * 1e-3),
                                     "frontend_TFLOPs": 8192 * 559 * (2 * 2 * 257 * 400) / (ms_fe * 1e-3) / 1e12}
    del big
    # ---- config 5 (FireRed half): 2048 x 10 s
    fr = firered.FireRedEngine(weights.firered_synthetic(1234))
    base = weights.burst_clips(32, 160000, seed=321)
    big = torch.from_numpy(base).cuda().repeat(64, 1)
    ms_fe = timed(torch, lambda: fr.fe.logmel(big, 10, 16000), args.reps)
    ms_all = timed(torch, lambda: fr.run(big, 10), args.reps)
    out["firered_config5"] = {"clips": 2048, "seconds_per_clip": 10, "ms_frontend": ms_fe, "ms_total": ms_all,
                              "hop512_frames_per_s": 2048 * 313 / (ms_all * 1e-3),
                              "net_TFLOPs": 2048 * 980 * 1.09e6 / ((ms_all - ms_fe) * 1e-3) / 1e12}
    # ---- config 5 (DFSMN near+far half): measured on 128 clip pairs x 10 s (1920 windows), scaled to 2048
    from vadx import dfsmn
    de = dfsmn.DfsmnEngine(weights.dfsmn_synthetic(1234), sub_batch=960)
    lb, stride = de.grid()
    W = 15
    n = (W - 1) * stride + de.L
    near = torch.from_numpy(weights.burst_clips(16, n, seed=11)).cuda().repeat(8, 1)
    far = torch.from_numpy(weights.burst_clips(16, n, seed=12)).cuda().repeat(8, 1)
    ms = timed(torch, lambda: de.run(near, far, W, stride), max(2, args.reps // 2))
    out["dfsmn_config5"] = {"clip_pairs_measured": 128, "windows": 128 * W, "ms_measured": ms,
                            "ms_scaled_to_2048_pairs": ms * 16, "hop512_frames_per_s": 128 * 313 / (ms * 1e-3),
                            "approx_TFLOPs": 128 * W * 4.8e9 / (ms * 1e-3) / 1e12}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
