"""CPU: the checkpoint / container readers of vadx.checkpoints and vadx.onnx_reader on SYNTHETIC containers written here
(torch zip pickles, a .nemo tar, Kaldi-nnet text, ONNX protobuf) -- the real files are not in the reference tree.  Each test
round-trips the seeded synthetic weight dict the GPU parity tests use: container -> loader -> identical arrays."""
import io
import os
import tarfile

import numpy as np
import pytest
import torch

import vadx  # noqa: F401
from vadx import checkpoints as ck
from vadx import onnx_reader as O

import _containers as CW           # test-only writers of the container formats (tests/_containers.py)
from vadx import weights


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def assert_same_dict(got, want, skip=()):
    assert set(got) - set(skip) == set(want) - set(skip), (sorted(set(got) ^ set(want)))
    for k in want:
        if k in skip:
            continue
        assert got[k].dtype == np.float32 and got[k].shape == np.asarray(want[k]).shape, k
        assert np.array_equal(got[k], want[k]), k


# ------------------------------------------------------------------ ONNX wire format
def test_onnx_reader_roundtrip_encodings_and_scopes(tmp_path):
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((4, 3, 3)).astype(np.float32), rng.standard_normal(4).astype(np.float32)
    h = rng.standard_normal((2, 5)).astype(np.float16)
    i64 = np.array([[1, -2], [3, 1 << 40]], np.int64)
    inner = CW.enc_graph([CW.enc_node("Conv", ["x", "w", "b"], ["y"], "/enc/conv", attrs={"kernel_shape": [3], "group": 1}),
                         CW.enc_node("Constant", [], ["/enc/Constant_output_0"], attrs={"value": h})],
                        [("w", a, True), ("b", b, False)], "then")                       # raw_data and packed float_data
    other = CW.enc_graph([], [("w", a * 2, True), ("shape", i64, False)], "else")          # packed int64_data
    top = CW.enc_graph([CW.enc_node("If", ["cond"], ["out"], "/If.1", attrs={"then_branch": ("graph", inner), "else_branch": ("graph", other)}),
                       CW.enc_node("Relu", ["out"], ["final"], attrs={"alpha": 0.25, "mode": b"x"})],
                      [("top.scale", np.array([1.5, -2.0], np.float64))])
    path = CW.write_onnx(str(tmp_path / "m.onnx"), top)
    g = O.read_onnx(path)
    then, els = (("/If.1", "then_branch"),), (("/If.1", "else_branch"),)
    assert g.scopes() == [then, els, ()]
    assert np.array_equal(g.tensors[(then, "w")], a) and np.array_equal(g.tensors[(then, "b")], b)
    assert g.tensors[(then, "/enc/Constant_output_0")].dtype == np.float16 and np.array_equal(g.tensors[(then, "/enc/Constant_output_0")], h)
    assert np.array_equal(g.tensors[(els, "w")], a * 2) and np.array_equal(g.tensors[(els, "shape")], i64)
    assert g.tensors[((), "top.scale")].dtype == np.float64
    vis = g.in_scope(then)
    assert set(vis) == {"top.scale", "w", "b", "/enc/Constant_output_0"} and np.array_equal(vis["w"], a)
    conv = [n for n in g.nodes if n.op_type == "Conv"][0]
    assert conv.inputs == ["x", "w", "b"] and conv.attrs["kernel_shape"] == [3] and conv.attrs["group"] == 1 and conv.scope == then
    relu = [n for n in g.nodes if n.op_type == "Relu"][0]
    assert abs(relu.attrs["alpha"] - 0.25) < 1e-7 and relu.attrs["mode"] == b"x"
    assert O.read_onnx(open(path, "rb").read()).tensors.keys() == g.tensors.keys()          # bytes in, same result
    with pytest.raises(ValueError):
        O.read_onnx(b"\x08\x08")                                                            # a ModelProto without a graph


def _silero_onnx(tmp_path, w, lstm_node, with_8k=True):
    """A Silero-v5-shaped container: top-level If(sr == 16000) -> 16 kHz sub-graph / 8 kHz sub-graph."""
    def branch(basis, conv0, tag):
        inits = [(f"{tag}.stft.forward_basis_buffer", basis)]
        nodes = [CW.enc_node("Conv", ["x", f"{tag}.stft.forward_basis_buffer"], ["spec"], f"/{tag}/stft/Conv", attrs={"strides": [128]})]
        prev = "mag"
        for i, cw in enumerate([conv0, w["enc1_w"], w["enc2_w"], w["enc3_w"]]):
            wn, bn = f"{tag}.encoder.{i}.reparam_conv.weight", f"onnx::Conv_{100 + i}_{tag}"     # bias name shares nothing with the weight
            inits += [(wn, cw, bool(i & 1)), (bn, w[f"enc{i}_b"], False)]
            nodes.append(CW.enc_node("Conv", [prev, wn, bn], [f"c{i}"], f"/{tag}/encoder.{i}/Conv"))
            prev = f"c{i}"
        H = 128
        if lstm_node:
            order = np.concatenate([np.arange(H) + H * gi for gi in (0, 3, 1, 2)])      # torch i,f,g,o -> ONNX i,o,f,c
            W, R = w["lstm_w_ih"][order][None], w["lstm_w_hh"][order][None]
            B = np.concatenate([w["lstm_b_ih"][order], w["lstm_b_hh"][order]])[None]
            inits += [(f"onnx::LSTM_{tag}_W", W), (f"onnx::LSTM_{tag}_R", R), (f"onnx::LSTM_{tag}_B", B)]
            nodes.append(CW.enc_node("LSTM", [prev, f"onnx::LSTM_{tag}_W", f"onnx::LSTM_{tag}_R", f"onnx::LSTM_{tag}_B", "", "h0", "c0"],
                                    ["y", "hn", "cn"], f"/{tag}/decoder/rnn/LSTM", attrs={"hidden_size": H}))
        else:
            inits += [(f"{tag}.decoder.rnn.weight_ih", w["lstm_w_ih"]), (f"{tag}.decoder.rnn.weight_hh", w["lstm_w_hh"]),
                      (f"{tag}.decoder.rnn.bias_ih", w["lstm_b_ih"]), (f"{tag}.decoder.rnn.bias_hh", w["lstm_b_hh"])]
        inits += [(f"{tag}.decoder.decoder.2.weight", w["dec_w"].reshape(1, 128, 1)), (f"{tag}.decoder.decoder.2.bias", w["dec_b"])]
        nodes.append(CW.enc_node("Conv", ["relu_h", f"{tag}.decoder.decoder.2.weight", f"{tag}.decoder.decoder.2.bias"], ["logit"], f"/{tag}/decoder/Conv"))
        return CW.enc_graph(nodes, inits, tag)

    rng = np.random.default_rng(8)
    g16 = branch(w["stft_basis"].reshape(258, 1, 256), w["enc0_w"], "m16")
    attrs = {"then_branch": ("graph", g16)}
    if with_8k:
        attrs["else_branch"] = ("graph", branch(rng.standard_normal((130, 1, 128)).astype(np.float32),
                                                rng.standard_normal((128, 65, 3)).astype(np.float32), "m8"))
    top = CW.enc_graph([CW.enc_node("Equal", ["sr", "c16k"], ["is16"]), CW.enc_node("If", ["is16"], ["out", "stateN"], "If_0", attrs=attrs)],
                      [("c16k", np.array(16000, np.int64))])
    return CW.write_onnx(str(tmp_path / f"silero_{int(lstm_node)}.onnx"), top)


@pytest.mark.parametrize("lstm_node", [True, False])
def test_silero_onnx_initialisers(tmp_path, lstm_node):
    w = weights.silero_synthetic(1234)
    path = _silero_onnx(tmp_path, w, lstm_node)
    got = ck.silero_from_onnx(path)
    assert_same_dict(got, w)
    assert_same_dict(ck.resolve("silero", path), w)
    weights.silero_check(got)
    with pytest.raises(ValueError, match="16 kHz sub-graph only"):
        ck.silero_from_onnx(path, sample_rate=8000)
    # a file without the 16 kHz tensors is refused with a listing, not half-loaded
    bad = CW.write_onnx(str(tmp_path / "bad.onnx"), CW.enc_graph([], [("x", np.zeros((3, 3, 3), np.float32))]))
    with pytest.raises(ValueError, match="weight shapes present"):
        ck.silero_from_onnx(bad)


# ------------------------------------------------------------------ FunASR FSMN: model.pt + am.mvn
def _write_am_mvn(path, means, scales):
    d = len(means)
    with open(path, "w") as fh:
        fh.write("<Nnet> \n<Splice> %d %d \n[ 0 ]\n<AddShift> %d %d \n" % (d, d, d, d))
        fh.write("<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in means) + " ]\n")
        fh.write("<Rescale> %d %d \n" % (d, d))
        fh.write("<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in scales) + " ]\n</Nnet> \n")


def test_funasr_fsmn_checkpoint(tmp_path):
    """State dict written from the REFERENCE's own encoder module names (FSMN/modeling_modified/encoder.py, the same mapping
    tests/golden/make_golden.py uses to load the synthetic weights into the reference net), `encoder.` prefix as FunASR saves it."""
    w = weights.fsmn_synthetic(7)
    sd = {"encoder.in_linear1.linear.weight": T(w["in1_w"]), "encoder.in_linear1.linear.bias": T(w["in1_b"]),
          "encoder.in_linear2.linear.weight": T(w["in2_w"]), "encoder.in_linear2.linear.bias": T(w["in2_b"]),
          "encoder.out_linear1.linear.weight": T(w["out1_w"]), "encoder.out_linear1.linear.bias": T(w["out1_b"]),
          "encoder.out_linear2.linear.weight": T(w["out2_w"]), "encoder.out_linear2.linear.bias": T(w["out2_b"]),
          "some.counter": torch.tensor(3)}
    for l in range(4):
        sd[f"encoder.fsmn.{l}.linear.linear.weight"] = T(w[f"l{l}_lin_w"])
        sd[f"encoder.fsmn.{l}.fsmn_block.conv_left.weight"] = T(w[f"l{l}_fir_w"]).reshape(128, 1, 20, 1)
        sd[f"encoder.fsmn.{l}.affine.linear.weight"] = T(w[f"l{l}_aff_w"])
        sd[f"encoder.fsmn.{l}.affine.linear.bias"] = T(w[f"l{l}_aff_b"])
    d = tmp_path / "fsmn"
    d.mkdir()
    torch.save(sd, str(d / "model.pt"))
    _write_am_mvn(str(d / "am.mvn"), w["cmvn_means"], w["cmvn_vars"])
    assert_same_dict(ck.load_fsmn(str(d)), w)
    assert_same_dict(ck.resolve("fsmn", str(d)), w)
    torch.save({"state_dict": sd}, str(d / "model.pt"))                       # wrapped the way trainer checkpoints are
    assert_same_dict(ck.load_fsmn(str(d / "model.pt")), w)
    sd5 = dict(sd)
    for k in list(sd):
        if ".fsmn.3." in k:
            sd5[k.replace(".fsmn.3.", ".fsmn.4.")] = sd[k]
    torch.save(sd5, str(d / "model.pt"))
    with pytest.raises(ValueError, match="5 memory blocks"):
        ck.load_fsmn(str(d))
    os.remove(str(d / "am.mvn"))
    with pytest.raises(ValueError, match="am.mvn"):
        ck.load_fsmn(str(d))


# ------------------------------------------------------------------ NeMo MarbleNet: .nemo
def _nemo_state(w):
    """NeMo ConvASREncoder layout: encoder.encoder.{b}.mconv = [dw MaskedConv1d, pw MaskedConv1d, BatchNorm1d, (ReLU, Dropout)] x repeat;
    res.0 = [1x1 MaskedConv1d, BatchNorm1d]; decoder = one Linear."""
    sd = {}
    for bi, (filt, rep, k, _s, _d, residual, sep) in enumerate(weights.MARBLENET_BLOCKS):
        idx = 0
        for r in range(rep):
            p = f"b{bi}r{r}"
            pre = f"encoder.encoder.{bi}.mconv."
            if sep:
                sd[f"{pre}{idx}.conv.weight"] = T(w[p + "_dw"]).unsqueeze(1)
                idx += 1
                sd[f"{pre}{idx}.conv.weight"] = T(w[p + "_pw"]).unsqueeze(-1)
            else:
                sd[f"{pre}{idx}.conv.weight"] = T(w[p + "_pw"]).unsqueeze(-1)
            idx += 1
            for src, dst in (("_gamma", "weight"), ("_beta", "bias"), ("_mean", "running_mean"), ("_var", "running_var")):
                sd[f"{pre}{idx}.{dst}"] = T(w[p + src])
            sd[f"{pre}{idx}.num_batches_tracked"] = torch.tensor(1000)
            idx += 1
            if r < rep - 1:
                idx += 2                                  # ReLU + Dropout sit in the list, parameter-free
        if residual:
            sd[f"encoder.encoder.{bi}.res.0.0.conv.weight"] = T(w[f"b{bi}res_pw"]).unsqueeze(-1)
            for src, dst in (("_gamma", "weight"), ("_beta", "bias"), ("_mean", "running_mean"), ("_var", "running_var")):
                sd[f"encoder.encoder.{bi}.res.0.1.{dst}"] = T(w[f"b{bi}res" + src])
    sd["decoder.layer0.weight"], sd["decoder.layer0.bias"] = T(w["dec_w"]), T(w["dec_b"])
    sd["preprocessor.featurizer.fb"] = torch.zeros(1, 80, 257)
    return sd


def test_nemo_marblenet_archive(tmp_path):
    w = weights.marblenet_synthetic(1234)
    sd = _nemo_state(w)
    buf = io.BytesIO()
    torch.save(sd, buf)
    cfg = "encoder:\n  jasper:\n" + "".join(
        f"  - filters: {f}\n    repeat: {r}\n    kernel: [{k}]\n    stride: [{s}]\n    dilation: [{d}]\n    residual: {str(res).lower()}\n    separable: {str(sep).lower()}\n"
        for f, r, k, s, d, res, sep in weights.MARBLENET_BLOCKS)
    path = str(tmp_path / "frame_vad.nemo")
    with tarfile.open(path, "w:gz") as tar:
        for name, blob in (("./model_config.yaml", cfg.encode()), ("./model_weights.ckpt", buf.getvalue())):
            info = tarfile.TarInfo(name)
            info.size = len(blob)
            tar.addfile(info, io.BytesIO(blob))
    got = ck.load_marblenet(path)
    assert_same_dict(got, w)
    assert_same_dict(ck.resolve("marblenet", path), w)
    # a config describing another Jasper stack is refused before any weight is mapped
    with tarfile.open(path, "w") as tar:
        for name, blob in (("model_config.yaml", cfg.replace("kernel: [13]", "kernel: [11]").encode()), ("model_weights.ckpt", buf.getvalue())):
            info = tarfile.TarInfo(name)
            info.size = len(blob)
            tar.addfile(info, io.BytesIO(blob))
    with pytest.raises(ValueError, match="model_config.yaml describes"):
        ck.load_marblenet(path)
    bad = dict(sd)
    bad["encoder.encoder.1.mconv.0.conv.weight"] = torch.zeros(128, 1, 11)
    with pytest.raises(ValueError, match="depthwise weight"):
        ck.marblenet_from_state(bad)


def test_nemo_loader_agrees_with_reference_bn_fold(golden):
    """The loader hands the engine unfolded conv + BatchNorm tensors; the engine folds with weights.fold_bn, which the
    marblenet_fold fixture pins against the reference's fold_bn_into_conv1d -- the chain archive -> folded weights is covered."""
    g = golden("marblenet_fold")
    for i in range(int(g["n_cases"])):
        b = g[f"b_{i}"] if bool(g[f"has_bias_{i}"]) else None
        fw, fb = weights.fold_bn(g[f"w_{i}"], b, g[f"gamma_{i}"], g[f"beta_{i}"], g[f"mean_{i}"], g[f"var_{i}"], float(g[f"eps_{i}"]))
        np.testing.assert_allclose(fw, g[f"fw_{i}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(fb, g[f"fb_{i}"], rtol=0, atol=1e-6)


# ------------------------------------------------------------------ resolve(): what a `weights` argument means
def test_resolve_never_defaults_to_random_weights(tmp_path):
    for kind in ("silero", "fsmn", "firered", "marblenet", "dfsmn"):
        for empty in (None, ""):
            with pytest.raises(ValueError, match="no weights given"):
                ck.resolve(kind, empty)
        w = ck.resolve(kind, "synthetic:7")                        # the explicit opt-in
        assert isinstance(w, dict) and len(w) > 4
        assert ck.resolve(kind, w) is w
        with pytest.raises(FileNotFoundError):
            ck.resolve(kind, str(tmp_path / "nope.bin"))
    p = str(tmp_path / "w.npz")
    np.savez(p, **weights.silero_synthetic(3))
    assert_same_dict({k: np.asarray(v, np.float32) for k, v in ck.resolve("silero", p).items()}, weights.silero_synthetic(3))
    jit = tmp_path / "silero_vad.jit"
    jit.write_bytes(b"PK")
    with pytest.raises(ValueError, match="neither a .onnx"):
        ck.resolve("silero", str(jit))


def test_readers_decode_hand_assembled_bytes():
    """Byte strings written out BY HAND from the format specifications (protobuf wire format + onnx.proto field numbers; Kaldi's
    binary matrix header), not produced by tests/_containers.py: a mistake shared by a reader and the test writers cannot pass."""
    # ModelProto { ir_version (1) = 8; graph (7) = GraphProto { name (2) = "g"; initializer (5) = TensorProto {
    #   dims (1) = 2, dims (1) = 2, data_type (2) = 1 (FLOAT), name (8) = "w", raw_data (9) = 4 little-endian floats } } }
    raw = np.array([1.0, -2.0, 0.5, 3.25], "<f4").tobytes()
    tensor = bytes([0x08, 2, 0x08, 2, 0x10, 1, 0x42, 1]) + b"w" + bytes([0x4A, 16]) + raw
    graph = bytes([0x12, 1]) + b"g" + bytes([0x2A, len(tensor)]) + tensor
    model = bytes([0x08, 8, 0x3A, len(graph)]) + graph
    g = O.read_onnx(model)
    assert list(g.tensors) == [((), "w")] and g.tensors[((), "w")].dtype == np.float32
    assert np.array_equal(g.tensors[((), "w")], np.array([[1.0, -2.0], [0.5, 3.25]], np.float32))
    # the same tensor with packed float_data (field 4) and int64 dims as multi-byte varints: dims = [1, 300] -> 300 = 0xAC 0x02
    vals = np.arange(300, dtype="<f4")
    packed = vals.tobytes()
    ln = len(packed)                                                    # 1200 = 0xB0 0x09
    tensor2 = bytes([0x08, 1, 0x08, 0xAC, 0x02, 0x10, 1, 0x22, 0xB0, 0x09]) + packed + bytes([0x42, 1]) + b"v"
    assert ln == 1200
    graph2 = bytes([0x2A]) + bytes([len(tensor2) & 0x7F | 0x80, len(tensor2) >> 7]) + tensor2
    g2 = O.read_onnx(bytes([0x3A, len(graph2) & 0x7F | 0x80, len(graph2) >> 7]) + graph2)
    assert g2.tensors[((), "v")].shape == (1, 300) and np.array_equal(g2.tensors[((), "v")][0], vals)
    # Kaldi binary matrix: "<key> " NUL 'B' "DM " then \4 int32 rows \4 int32 cols, then row-major little-endian doubles
    import struct
    import tempfile
    m = np.array([[1.5, -2.0, 3.0], [0.25, 8.0, -1.0]])
    blob = b"global \0BDM " + b"\x04" + struct.pack("<i", 2) + b"\x04" + struct.pack("<i", 3) + m.astype("<f8").tobytes()
    with tempfile.NamedTemporaryFile(suffix=".ark", delete=False) as fh:
        fh.write(blob)
    try:
        assert np.array_equal(ck.read_kaldi_matrix(fh.name), m)
        # float matrix ("FM ") and the text form "key  [\n rows ]"
        open(fh.name, "wb").write(b"k \0BFM " + b"\x04" + struct.pack("<i", 1) + b"\x04" + struct.pack("<i", 2) + np.array([7.0, -0.5], "<f4").tobytes())
        assert np.array_equal(ck.read_kaldi_matrix(fh.name), np.array([[7.0, -0.5]], np.float32))
        open(fh.name, "wb").write(b"global  [\n  1 2.5 3\n  4 5 6e1 ]\n")
        assert np.array_equal(ck.read_kaldi_matrix(fh.name), np.array([[1, 2.5, 3], [4, 5, 60.0]]))
    finally:
        os.unlink(fh.name)
