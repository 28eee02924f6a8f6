"""Real-checkpoint loaders: turn the files the reference's export scripts read into the weight dicts the
engines take, and `resolve()`, the one place that decides what an engine's `weights` argument means.

* FireRedVAD / AED / Stream-VAD  `model.pth.tar` + `cmvn.ark`   (FireRedVAD/Export_FireRedVAD.py:98-119, :328-364, :618-650)
* SDAEC ICCRN / alpha predictor  `ICCRN.ckpt`, `alpha.ckpt`      (DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py:362-366)
* FunASR FSMN-VAD                `model.pt` + `am.mvn`           (FSMN/Export_FSMN_VAD.py:107-113; layer names from the in-tree
                                                                  FSMN/modeling_modified/encoder.py)
* NeMo Frame-VAD MarbleNet       `*.nemo` (tar: model_weights.ckpt + model_config.yaml)
                                                                 (Export_NVIDIA_MarbleNet_VAD.py:360-362; module layout as
                                                                  fold_encoder_batchnorms walks it, :106-151)
* Silero                         `silero_vad.onnx` initialisers  (Silero/Export_Silero_VAD.py:91, utils_vad.py:117)

The first two container layouts are defined by code inside the reference tree.  The last three are written by third-party
packages (funasr, nemo, silero_vad) that are NOT in the reference tree and whose versions are unpinned: the readers below
decode the container FORMATS (torch zip pickles, tar, Kaldi-nnet text, ONNX protobuf) exactly, map tensors by the names
those packages are known to use AND by shape, and fail with a listing of what they found when a file does not match --
they are tested on synthetic containers written by the tests (tests/test_checkpoints.py), and are the only route by which
SURVEY rows a12 / a15 can be pinned on a machine that has the real files.
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


def _torch_load(src, what):
    """torch.load restricted to tensors / containers / plain namespaces (`weights_only=True`): a downloaded checkpoint is a
    pickle, and these loaders are reachable from any engine's `weights=<path>` string.  A file that needs more than that is
    refused unless the caller opts in with VADX_TRUST_CHECKPOINTS=1 (full unpickling executes code from the file: only for
    files whose origin is trusted)."""
    import argparse
    import pickle
    import types
    import torch
    try:
        with torch.serialization.safe_globals([argparse.Namespace, types.SimpleNamespace]):      # plain attribute holders (`args`)
            return torch.load(src, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if os.environ.get("VADX_TRUST_CHECKPOINTS") != "1":
            raise ValueError(f"{what}: the checkpoint holds objects beyond tensors / containers ({str(e).splitlines()[0]}); "
                             "set VADX_TRUST_CHECKPOINTS=1 to unpickle it fully if (and only if) you trust the file") from e
        if hasattr(src, "seek"):
            src.seek(0)
        return torch.load(src, map_location="cpu", weights_only=False)


def load_firered(model_dir):
    """`<model_dir>/model.pth.tar` + `<model_dir>/cmvn.ark` (the layout of the VAD / AED / Stream-VAD downloads,
    Export_FireRedVAD.py:13-15, :339-341) -> weight dict for FireRedEngine / FireRedStreamSession."""
    package = _torch_load(os.path.join(model_dir, "model.pth.tar"), "firered model.pth.tar")
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


# --------------------------------------------------------------------------- FunASR FSMN-VAD: model.pt + am.mvn
def read_kaldi_nnet_cmvn(path):
    """FunASR `am.mvn` (Kaldi nnet1 text: `<AddShift> d d` / `<LearnRateCoef> 0 [ ... ]`, `<Rescale> d d` / `<LearnRateCoef> 0 [ ... ]`)
    -> (means, vars) float32 [d]; the frontend applies (x + means) * vars, which is how the reference uses
    `frontend.cmvn[0]` / `[1]` (FSMN/Export_FSMN_VAD.py:85-86, :112-113)."""
    with open(path, "r", encoding="utf-8", errors="replace") as fh:
        lines = [ln.split() for ln in fh]
    means = scales = None
    for i, items in enumerate(lines):
        if not items:
            continue
        if items[0] in ("<AddShift>", "<Rescale>") and i + 1 < len(lines):
            row = lines[i + 1]
            if row and row[0] == "<LearnRateCoef>":
                lo, hi = row.index("["), (row.index("]") if "]" in row else len(row))
                vals = np.array([float(v) for v in row[lo + 1:hi]], dtype=np.float64)
                if items[0] == "<AddShift>":
                    means = vals
                else:
                    scales = vals
    if means is None or scales is None or means.shape != scales.shape:
        raise ValueError(f"{path}: no <AddShift> / <Rescale> pair found (not a FunASR am.mvn file?)")
    return means.astype(np.float32), scales.astype(np.float32)


_FSMN_KEYS = {"in_linear1.linear.weight": "in1_w", "in_linear1.linear.bias": "in1_b", "in_linear2.linear.weight": "in2_w",
              "in_linear2.linear.bias": "in2_b", "out_linear1.linear.weight": "out1_w", "out_linear1.linear.bias": "out1_b",
              "out_linear2.linear.weight": "out2_w", "out_linear2.linear.bias": "out2_b"}


def fsmn_from_state(state, cmvn):
    """state dict of the FunASR FSMN encoder (module names of FSMN/modeling_modified/encoder.py: in_linear1/2, fsmn.{l}.linear /
    .fsmn_block.conv_left / .affine, out_linear1/2; any prefix such as `encoder.` is stripped) + (means, vars) -> the weight
    dict of vadx.fsmn.FsmnEngine."""
    import torch
    t = {}
    for k, v in state.items():
        if not torch.is_tensor(v) or not v.is_floating_point():
            continue
        for marker in ("in_linear", "out_linear", "fsmn."):
            j = k.find(marker)
            if j >= 0 and (j == 0 or k[j - 1] == "."):
                t[k[j:]] = v.detach().to(torch.float32).numpy()
                break
    w = {}
    for src, dst in _FSMN_KEYS.items():
        if src not in t:
            raise ValueError(f"FSMN state dict lacks {src!r}; found {sorted(t)[:12]}")
        w[dst] = t[src]
    layer = 0
    while f"fsmn.{layer}.linear.linear.weight" in t:
        p = f"fsmn.{layer}."
        w[f"l{layer}_lin_w"] = t[p + "linear.linear.weight"]
        fir = t[p + "fsmn_block.conv_left.weight"]                 # Conv2d(P, P, [lorder, 1], groups=P): [P,1,lorder,1]
        w[f"l{layer}_fir_w"] = fir.reshape(fir.shape[0], -1)
        if p + "fsmn_block.conv_right.weight" in t:
            raise ValueError("FSMN checkpoint has a right-context filter (rorder > 0): not the streaming VAD encoder")
        w[f"l{layer}_aff_w"], w[f"l{layer}_aff_b"] = t[p + "affine.linear.weight"], t[p + "affine.linear.bias"]
        layer += 1
    if layer != 4:
        raise ValueError(f"FSMN checkpoint has {layer} memory blocks; the reference's cache layout fixes 4 (Export_FSMN_VAD.py:116-119)")
    means, scales = cmvn
    w["cmvn_means"], w["cmvn_vars"] = np.asarray(means, np.float32).reshape(-1), np.asarray(scales, np.float32).reshape(-1)
    if w["cmvn_means"].shape[0] != w["in1_w"].shape[1]:
        raise ValueError(f"CMVN dim {w['cmvn_means'].shape[0]} != encoder input dim {w['in1_w'].shape[1]}")
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def load_fsmn(model_dir):
    """`<model_dir>/model.pt` (+ `am.mvn`): the layout of the modelscope `speech_fsmn_vad_zh-cn-16k-common-pytorch` download
    that `AutoModel(model=model_path)` reads (FSMN/Export_FSMN_VAD.py:11, :107-113)."""
    import torch
    pt = os.path.join(model_dir, "model.pt") if os.path.isdir(model_dir) else model_dir
    pkg = _torch_load(pt, "fsmn model.pt")
    for key in ("state_dict", "model", "model_state_dict"):
        if isinstance(pkg, dict) and key in pkg and isinstance(pkg[key], dict):
            pkg = pkg[key]
    mvn = os.path.join(os.path.dirname(pt), "am.mvn")
    if not os.path.exists(mvn):
        raise ValueError(f"{mvn} is missing: the FSMN front-end CMVN (frontend.cmvn) comes from it")
    return fsmn_from_state(pkg, read_kaldi_nnet_cmvn(mvn))


# --------------------------------------------------------------------------- NeMo MarbleNet: *.nemo
def marblenet_from_state(state):
    """NeMo `EncDecFrameClassificationModel` state dict -> (weight dict of vadx.marblenet.MarbleNetEngine, blocks tuple).
    Layout (NeMo ConvASREncoder of JasperBlocks, walked the way the reference's fold_encoder_batchnorms does,
    Export_NVIDIA_MarbleNet_VAD.py:106-151): `encoder.encoder.{b}.mconv.{i}` is a ModuleList of MaskedConv1d (`.conv.weight`),
    BatchNorm1d (`.weight .bias .running_mean .running_var`) and parameter-free activations; a separable sub-block is
    depthwise [C,1,k] -> pointwise [C',C,1] -> BN; `encoder.encoder.{b}.res.0` = [1x1 conv, BN]; the decoder is one linear
    layer [2, 128] (any key under `decoder.`).  Kernel sizes / dilations that a weight shape cannot tell (stride, dilation)
    come from the published marblenet_3x2x64_20ms config (vadx.weights.MARBLENET_BLOCKS) and are checked against the shapes."""
    import re
    import torch
    from . import weights as _w
    t = {k: v.detach().to(torch.float32).numpy() for k, v in state.items() if torch.is_tensor(v) and v.is_floating_point()}
    blocks = {}
    for k in t:
        m = re.match(r"(?:.*\.)?encoder\.encoder\.(\d+)\.(mconv|res\.\d+)\.(\d+)\.(.+)$", k)
        if m:
            blocks.setdefault(int(m.group(1)), {}).setdefault(m.group(2), {}).setdefault(int(m.group(3)), {})[m.group(4)] = t[k]
    if not blocks:
        raise ValueError(f"no encoder.encoder.<b>.mconv.<i> keys; found {sorted(t)[:10]}")
    spec = _w.MARBLENET_BLOCKS
    if sorted(blocks) != list(range(len(spec))):
        raise ValueError(f"checkpoint has Jasper blocks {sorted(blocks)}; the published MarbleNet 3x2x64 has {len(spec)}")
    w = {}

    def bn(prefix, mod):
        try:
            w[prefix + "_gamma"], w[prefix + "_beta"] = mod["weight"], mod["bias"]
            w[prefix + "_mean"], w[prefix + "_var"] = mod["running_mean"], mod["running_var"]
        except KeyError as e:
            raise ValueError(f"{prefix}: BatchNorm tensors missing ({e})") from None

    cin = 80
    for bi, (filt, rep, k, _stride, _dil, residual, sep) in enumerate(spec):
        mods = [blocks[bi]["mconv"][i] for i in sorted(blocks[bi]["mconv"])]
        convs = [m for m in mods if "conv.weight" in m]
        norms = [m for m in mods if "running_mean" in m]
        if len(norms) != rep or len(convs) != rep * (2 if sep else 1):
            raise ValueError(f"block {bi}: {len(convs)} convs / {len(norms)} BatchNorms for repeat {rep}, separable {sep}")
        block_cin = cin
        for r in range(rep):
            p = f"b{bi}r{r}"
            if sep:
                dw, pw = convs[2 * r]["conv.weight"], convs[2 * r + 1]["conv.weight"]
                if dw.shape != (cin, 1, k):
                    raise ValueError(f"{p}: depthwise weight {dw.shape}, expected {(cin, 1, k)}")
                w[p + "_dw"] = dw[:, 0, :]
            else:
                pw = convs[r]["conv.weight"]
            if pw.shape[:2] != (filt, cin) or pw.shape[2] != (1 if sep else k):
                raise ValueError(f"{p}: pointwise weight {pw.shape}, expected {(filt, cin, 1 if sep else k)}")
            if any("conv.bias" in c for c in convs):
                raise ValueError(f"{p}: conv bias present; the Jasper convs of this model have none")
            w[p + "_pw"] = pw[:, :, 0]
            bn(p, norms[r])
            cin = filt
        if residual:
            res = blocks[bi].get("res.0")
            if not res:
                raise ValueError(f"block {bi}: residual branch res.0 missing")
            rmods = [res[i] for i in sorted(res)]
            rc = [m for m in rmods if "conv.weight" in m][0]["conv.weight"]
            if rc.shape != (filt, block_cin, 1):
                raise ValueError(f"block {bi}: residual conv {rc.shape}, expected {(filt, block_cin, 1)}")
            w[f"b{bi}res_pw"] = rc[:, :, 0]
            bn(f"b{bi}res", [m for m in rmods if "running_mean" in m][0])
    dec_w = [v for k2, v in t.items() if "decoder." in k2 and k2.endswith("weight") and v.shape[0] == 2 and v.size == 2 * cin]
    dec_b = [v for k2, v in t.items() if "decoder." in k2 and k2.endswith("bias") and v.shape == (2,)]
    if len(dec_w) != 1 or len(dec_b) != 1:
        raise ValueError(f"decoder: expected one [2,{cin}] weight and one [2] bias under 'decoder.', found {len(dec_w)} / {len(dec_b)}")
    w["dec_w"], w["dec_b"] = dec_w[0].reshape(2, cin), dec_b[0]
    return {k2: np.ascontiguousarray(v, dtype=np.float32) for k2, v in w.items()}


def load_marblenet(nemo_path):
    """`*.nemo` = a (possibly gzipped) tar holding `model_weights.ckpt` (a torch state dict) and `model_config.yaml`
    (what `EncDecSpeakerLabelModel.restore_from` unpacks, Export_NVIDIA_MarbleNet_VAD.py:360-362) -> MarbleNetEngine weights
    (conv + BatchNorm statistics; the engine folds them like the reference's fold_bn_into_conv1d)."""
    import io
    import tarfile
    import torch
    with tarfile.open(nemo_path, "r:*") as tar:
        member = next((m for m in tar.getmembers() if os.path.basename(m.name) == "model_weights.ckpt"), None)
        if member is None:
            raise ValueError(f"{nemo_path}: no model_weights.ckpt inside (members: {[m.name for m in tar.getmembers()][:8]})")
        blob = tar.extractfile(member).read()
        cfg_m = next((m for m in tar.getmembers() if os.path.basename(m.name) == "model_config.yaml"), None)
        if cfg_m is not None:
            import yaml
            cfg = yaml.safe_load(tar.extractfile(cfg_m).read()) or {}
            jasper = (cfg.get("encoder") or {}).get("jasper")
            if jasper:
                from . import weights as _w
                got = tuple((int(j["filters"]), int(j["repeat"]), int(j["kernel"][0]), int(j["stride"][0]), int(j["dilation"][0]),
                             bool(j.get("residual", False)), bool(j.get("separable", False))) for j in jasper)
                if got != tuple(_w.MARBLENET_BLOCKS):
                    raise ValueError(f"model_config.yaml describes {got}; the HIP net is built for {_w.MARBLENET_BLOCKS}")
    state = _torch_load(io.BytesIO(blob), "marblenet model_weights.ckpt")
    if isinstance(state, dict) and "state_dict" in state:
        state = state["state_dict"]
    return marblenet_from_state(state)


# --------------------------------------------------------------------------- Silero: silero_vad.onnx
_ONNX_TO_TORCH_GATES = (0, 2, 3, 1)          # ONNX LSTM packs gates i, o, f, c; torch.nn.LSTMCell (and the HIP kernel) i, f, g, o


def silero_from_onnx(path_or_bytes, sample_rate=16000):
    """Initialisers of a Silero-VAD v5 `.onnx` -> the weight dict of vadx.silero.SileroEngine.

    The file (silero_vad pip package, not in the reference tree) is one graph whose top-level `If` selects a 16 kHz or an
    8 kHz sub-graph; each holds the STFT conv basis, four Conv(k=3)+ReLU encoder layers, one LSTM cell (as an ONNX `LSTM` node with
    W [1,4H,I] / R [1,4H,H] / B [1,8H] in i-o-f-c gate order, or as plain [4H,I] / [4H,H] matrices named weight_ih / weight_hh
    in torch order) and a 1x1 output conv.  Tensors are picked per scope by SHAPE (16 kHz: basis [258,1,256], convs
    [128,129,3] [64,128,3] [64,64,3] [128,64,3]), biases through the Conv node that consumes each weight; names are used only
    to tell weight_ih from weight_hh.  Raises with the scope's tensor listing when the file does not look like that."""
    from . import onnx_reader
    g = onnx_reader.read_onnx(path_or_bytes)
    if sample_rate != 16000:
        raise ValueError("the HIP Silero network implements the 16 kHz sub-graph only (the 8 kHz one has a 128-point STFT and a "
                         "65-channel first conv); use sampling_rate=16000")
    want = {"stft_basis": (258, 1, 256), "enc0_w": (128, 129, 3), "enc1_w": (64, 128, 3), "enc2_w": (64, 64, 3), "enc3_w": (128, 64, 3)}
    chosen = None
    for sc in g.scopes():
        vis = g.in_scope(sc)
        if all(any(v.shape == shp for v in vis.values()) for shp in want.values()):
            chosen = sc if chosen is None or len(sc) > len(chosen) else chosen
    if chosen is None:
        listing = sorted({(tuple(v.shape)) for v in g.tensors.values() if v.ndim >= 2})
        raise ValueError(f"no (sub)graph holds the 16 kHz Silero tensors {sorted(want.values())}; weight shapes present: {listing[:24]}")
    vis = g.in_scope(chosen)
    nodes = [n for n in g.nodes if n.scope == chosen[:len(n.scope)] and len(n.scope) <= len(chosen)]

    def by_shape(shape):
        hits = [k for k, v in vis.items() if v.shape == shape]
        if len(hits) != 1:
            raise ValueError(f"{len(hits)} tensors of shape {shape} in the chosen sub-graph ({hits}); cannot map the Silero weights")
        return hits[0]

    def bias_of(wname, n_out):
        for n in nodes:
            if n.op_type == "Conv" and len(n.inputs) >= 3 and n.inputs[1] == wname and n.inputs[2] in vis:
                return vis[n.inputs[2]].reshape(-1)
        # no node list to go by (weights only): the bias is the 1-D tensor whose name shares the weight's stem
        stem = wname.rsplit("weight", 1)[0]
        hits = [v for k, v in vis.items() if v.ndim == 1 and v.shape[0] == n_out and k.startswith(stem) and "bias" in k]
        if len(hits) == 1:
            return hits[0]
        raise ValueError(f"cannot find the bias of conv weight {wname!r}")

    w = {"stft_basis": vis[by_shape(want["stft_basis"])].reshape(258, 256)}
    for i in range(4):
        name = by_shape(want[f"enc{i}_w"])
        w[f"enc{i}_w"] = vis[name]
        w[f"enc{i}_b"] = bias_of(name, want[f"enc{i}_w"][0])
    H = 128
    lstm = [n for n in nodes if n.op_type == "LSTM"]
    if lstm:
        n = lstm[0]
        W, R, B = (vis[n.inputs[i]] for i in (1, 2, 3))
        if W.shape != (1, 4 * H, H) or R.shape != (1, 4 * H, H) or B.shape != (1, 8 * H):
            raise ValueError(f"LSTM node tensors {W.shape} {R.shape} {B.shape}, expected (1,512,128) (1,512,128) (1,1024)")
        order = np.concatenate([np.arange(H) + H * gidx for gidx in _ONNX_TO_TORCH_GATES])
        w["lstm_w_ih"], w["lstm_w_hh"] = W[0][order], R[0][order]
        w["lstm_b_ih"], w["lstm_b_hh"] = B[0, :4 * H][order], B[0, 4 * H:][order]
    else:
        def named(sub, shape):
            hits = [k for k, v in vis.items() if v.shape == shape and sub in k]
            if len(hits) != 1:
                raise ValueError(f"expected one tensor of shape {shape} with {sub!r} in its name, found {hits}")
            return vis[hits[0]]
        w["lstm_w_ih"], w["lstm_w_hh"] = named("weight_ih", (4 * H, H)), named("weight_hh", (4 * H, H))
        w["lstm_b_ih"], w["lstm_b_hh"] = named("bias_ih", (4 * H,)), named("bias_hh", (4 * H,))
    dname = by_shape((1, H, 1))
    w["dec_w"], w["dec_b"] = vis[dname].reshape(H), bias_of(dname, 1).reshape(1)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


# --------------------------------------------------------------------------- what an engine's `weights` argument means
_SYNTH = {"silero": "silero_synthetic", "fsmn": "fsmn_synthetic", "firered": "firered_synthetic",
          "marblenet": "marblenet_synthetic", "dfsmn": "dfsmn_synthetic"}


def resolve(kind, spec):
    """dict -> used as is; "synthetic:<seed>" -> seeded random weights of the real architecture (an EXPLICIT opt-in: scores
    are meaningless, only shapes / timings / parity tests want them); a path -> the loader for `kind`; None / "" -> ValueError.
    The reference's defaults (packaged silero_vad.onnx, modelscope / NGC downloads) do not exist here, and silently
    substituting random weights would hand a drop-in caller plausible-looking garbage."""
    from . import weights as _w
    if isinstance(spec, dict):
        return spec
    if spec is None or spec == "":
        raise ValueError(f"{kind}: no weights given.  Pass a weight dict, a checkpoint path "
                         f"({_HINT[kind]}), or the explicit opt-in 'synthetic:<seed>' for seeded random weights.")
    s = str(spec)
    if s.startswith("synthetic"):
        seed = int(s.split(":", 1)[1]) if ":" in s else 1234
        return getattr(_w, _SYNTH[kind])(seed)
    if not os.path.exists(s):
        raise FileNotFoundError(f"{kind}: weights path {s!r} does not exist")
    low = s.lower()
    if low.endswith(".npz"):
        with np.load(s) as z:
            return {k: z[k] for k in z.files}
    if kind == "silero":
        if low.endswith(".onnx"):
            return silero_from_onnx(s)
        raise ValueError(f"silero: {s!r} is neither a .onnx file nor a .npz of arrays (a TorchScript .jit cannot be read without its code)")
    if kind == "fsmn":
        return load_fsmn(s)
    if kind == "firered":
        return load_firered(s)
    if kind == "marblenet":
        if low.endswith(".nemo"):
            return load_marblenet(s)
        raise ValueError(f"marblenet: {s!r} is not a .nemo archive")
    if kind == "dfsmn":
        raise ValueError("dfsmn: build the dict from load_dfsmn_aec(<SDAEC dir>) plus the modelscope mask-net arrays ('mask.*'); "
                         "there is no single-file checkpoint for this model")
    raise ValueError(f"unknown model kind {kind!r}")


_HINT = {"silero": "silero_vad.onnx or .npz", "fsmn": "FunASR model dir with model.pt + am.mvn", "firered": "dir with model.pth.tar + cmvn.ark",
         "marblenet": "frame_vad_multilingual_marblenet_v2.0.nemo", "dfsmn": "dict from load_dfsmn_aec + mask-net arrays"}
