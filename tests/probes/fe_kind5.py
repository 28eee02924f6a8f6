"""log-mel error of every front-end product kind on the one case where kind 5 exceeds FEAT_ATOL (test_logmel_matches_oracle[5-fsmn-16000-3-11040-3])"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import vadx
from vadx import frontend, weights
from test_gpu_frontend import oracle_logmel, T
preset, L, W, stride, B = "fsmn", 16000, 3, 11040, 3
n = (W - 1) * stride + L
clips = weights.burst_clips(B, n, seed=L + W + B)
clips[0, :3000] = 0
wins = np.stack([clips[b, w * stride:w * stride + L] for b in range(B) for w in range(W)])
ref = oracle_logmel(preset, T(wins).unsqueeze(1)).numpy()
ref64 = None
big = ref > np.log(1e-3)
for fold in (False, True, 4, 5):
    fe = frontend.Frontend(preset, L, fold=fold)
    out = fe.logmel(clips, windows_per_clip=W, win_stride=stride).cpu().numpy()
    err = np.abs(out - ref)
    e = err[big]
    idx = np.unravel_index(np.argmax(np.where(big, err, 0)), err.shape)
    print(f"kind {fe.fold}: max err above the floor {e.max():.3e} at {idx} (ref {ref[idx]:.3f}, frame max {ref[idx[0], idx[1]].max():.3f}); mean {e.mean():.3e}; "
          f"entries > 1e-4: {(e > 1e-4).sum()}, > 1.5e-4: {(e > 1.5e-4).sum()}, > 2e-4: {(e > 2e-4).sum()} of {e.size}")
