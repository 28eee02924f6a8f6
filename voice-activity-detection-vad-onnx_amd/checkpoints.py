"""Real-checkpoint loaders: turn the files the reference's export scripts read into the weight dicts the
engines take.  Only what the reference itself specifies is implemented:

* FireRedVAD / AED / Stream-VAD  `model.pth.tar` + `cmvn.ark`   (FireRedVAD/Export_FireRedVAD.py:98-119, :328-364, :618-650)
* SDAEC ICCRN / alpha predictor  `ICCRN.ckpt`, `alpha.ckpt`      (DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py:362-366)

The FunASR FSMN (`model.pt` + `am.mvn`), NeMo MarbleNet (`.nemo`) and Silero (`.onnx`) containers are read by third-party
packages that are not part of the reference tree (funasr, nemo, onnxruntime); their key layouts cannot be pinned here,
so those engines take a plain dict / `.npz` of arrays (see each engine's docstring for the expected names).
"""
from __future__ import annotations

import math
import os
import struct

import numpy as np


# --------------------------------------------------------------------------- Kaldi matrices (what kaldiio.load_mat reads)
def read_kaldi_matrix(path):
    """Kaldi matrix from an ark / mat file, binary (`[key ]\\0B{F,D}M \\4rows\\4cols data`) or text (`[key ] [ .. \\n .. ]`);
    `path` may carry an `:offset` suffix as in scp entries.  -> float32/float64 numpy [rows, cols]."""
    offset = 0
    if ":" in os.path.basename(path) and not os.path.exists(path):
        path, off = path.rsplit(":", 1)
        offset = int(off)
    with open(path, "rb") as fh:
        buf = fh.read()
    pos = offset
    if offset == 0:                                  # optional utterance key up to the first space
        mark = buf.find(b"\0B")
        br = buf.find(b"[")
        if mark >= 0 and (br < 0 or mark < br):
            pos = mark
        elif br >= 0:
            pos = br
    if buf[pos:pos + 2] == b"\0B":
        pos += 2
        tok = buf[pos:pos + 3]
        if tok not in (b"FM ", b"DM "):
            raise ValueError(f"{path}: unsupported Kaldi binary type {tok!r} (only FM / DM matrices)")
        pos += 3
        dims = []
        for _ in range(2):
            if buf[pos] != 4:
                raise ValueError(f"{path}: malformed Kaldi matrix header")
            dims.append(struct.unpack_from("<i", buf, pos + 1)[0])
            pos += 5
        dt = np.float32 if tok == b"FM " else np.float64
        n = dims[0] * dims[1]
        return np.frombuffer(buf, dtype=dt, count=n, offset=pos).reshape(dims).copy()
    text = buf[pos:].decode("ascii", "replace")
    lo, hi = text.index("["), text.index("]")
    rows = [r.split() for r in text[lo + 1:hi].strip().split("\n") if r.strip()]
    return np.array(rows, dtype=np.float64)


def write_kaldi_matrix(path, mat, key="global", binary=True):
    """Inverse of read_kaldi_matrix (used by the tests and for exporting statistics)."""
    m = np.ascontiguousarray(mat)
    with open(path, "wb") as fh:
        if binary:
            tok = b"DM " if m.dtype == np.float64 else b"FM "
            if tok == b"FM ":
                m = m.astype(np.float32)
            fh.write(key.encode() + b" \0B" + tok + b"\4" + struct.pack("<i", m.shape[0]) + b"\4" + struct.pack("<i", m.shape[1]))
            fh.write(m.tobytes())
        else:
            body = "\n".join("  " + " ".join(repr(float(v)) for v in row) for row in m)
            fh.write((key + "  [\n" + body + " ]\n").encode())


def load_cmvn(cmvn_file):
    """(means, inverse standard deviations) float32 [dim] from Kaldi CMVN statistics [2, dim+1], with the float32
    arithmetic of the reference (FireRedVAD/Export_FireRedVAD.py:98-119)."""
    stats = np.asarray(read_kaldi_matrix(cmvn_file), dtype=np.float32)
    if stats.shape[0] != 2:
        raise ValueError("CMVN statistics must have two rows (sums, sums of squares)")
    dim = stats.shape[1] - 1
    count = stats[0, dim]
    if not count >= 1:
        raise ValueError("CMVN statistics: count < 1")
    means = np.zeros(dim, dtype=np.float32)
    inv_std = np.zeros(dim, dtype=np.float32)
    for d in range(dim):
        mean = np.float32(stats[0, d] / count)
        means[d] = mean
        variance = np.float32(np.float32(stats[1, d] / count) - np.float32(mean * mean))
        if variance < np.float32(1e-20):
            variance = np.float32(1e-20)
        inv_std[d] = np.float32(1.0 / math.sqrt(float(variance)))
    return means, inv_std


# --------------------------------------------------------------------------- FireRed
def firered_from_state(args, state, cmvn=None):
    """`package["args"]` (attributes idim R M H P N1 S1 N2 S2 odim) + `package["model_state_dict"]` (Linear weights 2-D,
    FIR filters [P,1,N]) -> the weight dict of `vadx.firered.FireRedEngine`, CMVN fused into fc1 exactly as
    DetectModel.from_pretrained does (Export_FireRedVAD.py:350-360).  A streaming checkpoint simply has no look-ahead
    filters (N2 = 0)."""
    import torch
    get = (lambda k: args[k]) if isinstance(args, dict) else (lambda k: getattr(args, k))
    has_la = any("lookahead_filter" in k for k in state)
    cfg = {k: int(get(k)) for k in ("idim", "R", "M", "H", "P", "N1", "S1", "odim")}
    cfg["N2"] = int(get("N2")) if has_la else 0
    cfg["S2"] = int(get("S2")) if has_la else 0
    t = {k: (v if torch.is_tensor(v) else torch.as_tensor(v)).detach().to(torch.float32) for k, v in state.items()}
    W, b = t["dfsmn.fc1.0.weight"].reshape(cfg["H"], cfg["idim"]), t["dfsmn.fc1.0.bias"]
    if cmvn is not None:
        means, inv_std = (torch.as_tensor(np.asarray(c, dtype=np.float32)) for c in cmvn)
        b = b - torch.mv(W, means * inv_std)
        W = W * inv_std.view(1, -1)
    w = {"cfg": cfg, "fc1_w": W, "fc1_b": b,
         "fc2_w": t["dfsmn.fc2.0.weight"].reshape(cfg["P"], cfg["H"]), "fc2_b": t["dfsmn.fc2.0.bias"],
         "fsmn0_lb": t["dfsmn.fsmn1.lookback_filter.weight"].reshape(cfg["P"], cfg["N1"]),
         "out_w": t["out.weight"].reshape(cfg["odim"], cfg["H"]), "out_b": t["out.bias"]}
    if has_la:
        w["fsmn0_la"] = t["dfsmn.fsmn1.lookahead_filter.weight"].reshape(cfg["P"], cfg["N2"])
    for r in range(1, cfg["R"]):
        p = f"dfsmn.fsmns.{r - 1}."
        w[f"blk{r}_fc1_w"] = t[p + "fc1.0.weight"].reshape(cfg["H"], cfg["P"])
        w[f"blk{r}_fc1_b"] = t[p + "fc1.0.bias"]
        w[f"blk{r}_fc2_w"] = t[p + "fc2.weight"].reshape(cfg["P"], cfg["H"])
        w[f"fsmn{r}_lb"] = t[p + "fsmn.lookback_filter.weight"].reshape(cfg["P"], cfg["N1"])
        if has_la:
            w[f"fsmn{r}_la"] = t[p + "fsmn.lookahead_filter.weight"].reshape(cfg["P"], cfg["N2"])
    for m in range(cfg["M"]):
        w[f"dnn{m}_w"] = t[f"dfsmn.dnns.{2 * m}.weight"].reshape(cfg["H"], cfg["P"] if m == 0 else cfg["H"])
        w[f"dnn{m}_b"] = t[f"dfsmn.dnns.{2 * m}.bias"]
    return {k: (v if k == "cfg" else np.ascontiguousarray(v.numpy(), dtype=np.float32)) for k, v in w.items()}


def load_firered(model_dir):
    """`<model_dir>/model.pth.tar` + `<model_dir>/cmvn.ark` (the layout of the VAD / AED / Stream-VAD downloads,
    Export_FireRedVAD.py:13-15, :339-341) -> weight dict for FireRedEngine / FireRedStreamSession."""
    import torch
    package = torch.load(os.path.join(model_dir, "model.pth.tar"), map_location="cpu", weights_only=False)
    cmvn_path = os.path.join(model_dir, "cmvn.ark")
    cmvn = load_cmvn(cmvn_path) if os.path.exists(cmvn_path) else None
    return firered_from_state(package["args"], package["model_state_dict"], cmvn)


# --------------------------------------------------------------------------- SDAEC ICCRN + alpha predictor
def load_dfsmn_aec(model_dir):
    """`<model_dir>/ICCRN.ckpt` and `alpha.ckpt` are plain state dicts of NET / AlphaPredictor
    (Export_DFSMN_VAD.py:362-366) -> {'iccrn.<name>': array, 'alpha.<name>': array}; merge with the mask-net arrays
    ('mask.<name>', from the modelscope pipeline's model) to build a DfsmnEngine."""
    import torch
    out = {}
    for fname, prefix in (("ICCRN.ckpt", "iccrn."), ("alpha.ckpt", "alpha.")):
        sd = torch.load(os.path.join(model_dir, fname), map_location="cpu")
        for k, v in sd.items():
            if torch.is_tensor(v) and v.is_floating_point():
                out[prefix + k] = np.ascontiguousarray(v.detach().to(torch.float32).numpy())
    return out
