"""CPU: the C-ABI library builds, loads and exports every symbol include/vadx.h declares
(no compute calls -- there is no GPU here), and host-only logic of the product package."""
import ctypes
import os
import re

import numpy as np
import pytest

import vadx  # noqa: F401
from vadx import _lib, build, timestamps, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "vadx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vadx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libvadx.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.vadx_abi_version() == _lib.ABI_VERSION


def test_graft_entry_abi_check(lib):
    """One number, one place: include/vadx.h's VADX_ABI_VERSION is what capi.hip returns, what vadx._lib expects, what the C
    client compares with and what __graft_entry__.build() asserts (round 3 ended with that last one stale and build() raising)."""
    import __graft_entry__ as g
    hdr = open(os.path.join(ROOT, "include", "vadx.h")).read()
    n = int(re.search(r"^#define\s+VADX_ABI_VERSION\s+(\d+)", hdr, flags=re.M).group(1))
    assert _lib.ABI_VERSION == n
    assert g.check_abi(lib) == n
    for rel in ("voice-activity-detection-vad-onnx_amd/csrc/capi.hip", "tests/c/cabi_silero.c", "__graft_entry__.py"):
        src = open(os.path.join(ROOT, rel)).read()
        assert not re.search(r"abi_version\(\)\s*[!=]=\s*\d", src), f"{rel} compares the ABI with a literal"
        assert not re.search(r"vadx_abi_version\(void\)\s*\{\s*return\s+\d", src), f"{rel} returns a literal ABI number"


def test_pack_host_layout(lib):
    """Host-only repack (no GPU): spot-check the packed blob against the documented layout."""
    w = weights.silero_synthetic(3)
    hw = _lib.SileroWeightsHost()
    hw.stft_basis = w["stft_basis"].ctypes.data
    for i in range(4):
        hw.enc_w[i] = w[f"enc{i}_w"].ctypes.data
        hw.enc_b[i] = w[f"enc{i}_b"].ctypes.data
    for k in ("lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh", "dec_w", "dec_b"):
        setattr(hw, k, w[k].ctypes.data)
    n = lib.vadx_silero_packed_floats()
    p = np.zeros(n, np.float32)
    assert lib.vadx_silero_pack_host(ctypes.byref(hw), p.ctypes.data) == 0
    # STFT rows regrouped per wave: [wave][re|im][16][256]
    stft = p[:256 * 256].reshape(8, 2, 16, 256)
    perm = np.array([16 * S + q + 4 * j for S in range(16) for q in range(4) for j in range(4)])   # slot 16S+4q+j <- k
    assert np.array_equal(stft[3, 0, 5], w["stft_basis"][3 * 16 + 5][perm])
    assert np.array_equal(stft[3, 1, 5], w["stft_basis"][129 + 3 * 16 + 5][perm])
    nyq = p[65536:65536 + 512].reshape(2, 256)
    assert np.array_equal(nyq[0], w["stft_basis"][128]) and np.array_equal(nyq[1], w["stft_basis"][257])
    # conv1 lives in the Winograd F(4,3) domain: six planes U_j = G g (float64 on the host), input channels 0..127
    # fragment-major [8 oc tiles][6 planes][8 blocks][64 lanes][4], input channel 128 as [128 oc][6 (+2 pad)]
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
    U = np.einsum("jt,oct->ojc", G, w["enc0_w"].astype(np.float64)).astype(np.float32)      # [oc][plane][ch]
    c1 = p[66048:66048 + 128 * 6 * 128].reshape(8, 6, 8, 4, 16, 4)          # tile, plane, S, q, i, j
    rows = c1.transpose(0, 4, 1, 2, 3, 5).reshape(128, 6, 128)              # -> [oc][plane][k = 16S + 4q + j]
    np.testing.assert_allclose(rows, U[:, :, :128], rtol=2e-7, atol=0)        # float64 sums, one float32 rounding (summation order may differ)
    assert np.array_equal(rows[:, 5], w["enc0_w"][:, :128, 2]) and np.array_equal(rows[:, 0], w["enc0_w"][:, :128, 0] * np.float32(0.25))
    c1n = p[66048 + 128 * 6 * 128:66048 + 128 * 6 * 128 + 1024].reshape(128, 8)
    np.testing.assert_allclose(c1n[:, :6], U[:, :, 128], rtol=2e-7, atol=0)
    assert not c1n[:, 6:].any()
    # the Winograd identity itself, in float64: A^T [(G g) . (B^T d)] == direct conv over frames -1..4 with zero frames at the edges
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
    rng = np.random.default_rng(0)
    d = np.zeros((6, 129))
    d[1:5] = rng.uniform(0, 3, (4, 129))
    g = w["enc0_w"].astype(np.float64)
    direct = np.stack([sum(g[:, :, t] @ d[f + t] for t in range(3)) for f in range(4)])
    wino = AT @ np.stack([np.einsum("oc,c->o", np.einsum("t,oct->oc", G[j], g), BT[j] @ d) for j in range(6)])
    np.testing.assert_allclose(wino, direct, rtol=0, atol=1e-12)
    # the generic helper produces the same layout
    a = np.arange(20 * 37, dtype=np.float32).reshape(20, 37)
    fm = _lib.frag_major(a).reshape(2, 3, 4, 16, 4)                         # tile, S, q, i, j  (32 x 48 padded)
    back = fm.transpose(0, 3, 1, 2, 4).reshape(32, 48)
    assert np.array_equal(back[:20, :37], a) and not back[20:].any() and not back[:, 37:].any()
    # NULL pointer -> EINVAL with a message
    assert lib.vadx_silero_pack_host(None, p.ctypes.data) == -1
    assert b"NULL" in lib.vadx_last_error()


def test_fp16x2_fragment_packer_layout_and_range():
    """vadx_frag_h2_host: the two round-to-nearest fp16 terms of every weight land in lane (q, i) slot e of their (tile, chunk) fragment pair with
    k = 32 chunk + 16 (e >> 2) + 4 q + (e & 3); h0 + h1 2^-11 reproduces the weight to a float32 ulp; a weight outside the fp16 range is refused."""
    from vadx import _lib
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((20, 70)) * np.exp(rng.uniform(-6, 6, (20, 70)))).astype(np.float32)
    fr = _lib.frag_h2(a, _lib.H2_K_QUARTER)
    assert fr is not None and fr.size == 2 * 3 * 2 * 256
    plain = _lib.frag_h2(a).view(np.float16).reshape(2, 3, 2, 4, 16, 8)     # VADX_H2_K_PLAIN: k = 8 q + e
    h = fr.view(np.float16).reshape(2, 3, 2, 4, 16, 8)                      # tile, chunk, plane, q, i, e
    back = np.zeros((32, 96), np.float64)
    for q in range(4):
        for e in range(8):
            k = 16 * (e >> 2) + 4 * q + (e & 3)
            for c in range(3):
                back[:, 32 * c + k] = (h[:, c, 0, q, :, e].astype(np.float64) + h[:, c, 1, q, :, e].astype(np.float64) / 2048.0).reshape(32)
    assert not back[20:].any() and not back[:, 70:].any()
    pb = (plain[:, :, 0].astype(np.float64) + plain[:, :, 1].astype(np.float64) / 2048.0).transpose(0, 3, 1, 2, 4).reshape(32, 96)
    assert np.array_equal(pb, back)
    err = np.abs(back[:20, :70] - a.astype(np.float64))
    assert np.all(err <= np.abs(a) * 2.0 ** -23 + 2.0 ** -36), float((err / np.abs(a)).max())
    a[3, 5] = 7e4
    assert _lib.frag_h2(a) is None and b"fp16 range" in _lib.lib().vadx_last_error()


def test_dfsmn_entry_points_validate_before_touching_the_device():
    """Argument checks of the ICCRN building blocks run on the host, before any HIP call: unsupported shapes / missing operands
    come back as VADX_EINVAL with a message (the pw_conv kernel is instantiated for the ICCRN's seven (co, cin, kf, mode) only)."""
    import ctypes as C
    lib = _lib.lib()
    buf = (C.c_float * 64)()
    ptr = C.cast(buf, C.c_void_p).value
    v = lambda ct, c: _lib.FtView(ptr, ct, 0, c)                                    # noqa: E731
    ln = _lib.FtLn(ptr, ptr, ptr)
    pw = lambda mode, a, lnp, co, kf: lib.vadx_dfsmn_pw_conv(mode, C.byref(a), None, lnp, ptr, ptr, None, None, None, C.byref(v(20, 20)),   # noqa: E731
                                                             None, 160, co, kf, 0, 4, None, None, None)
    assert pw(0, v(20, 20), None, 7, 1) == -1 and b"unsupported shape" in lib.vadx_last_error()
    assert pw(0, v(20, 20), C.byref(ln), 20, 1) == -1 and b"LayerNorm" in lib.vadx_last_error()          # mode 0 takes none
    assert pw(2, v(20, 20), None, 20, 3) == -1 and b"LayerNorm" in lib.vadx_last_error()                   # mode 2 needs one
    assert pw(0, v(22, 22), None, 20, 1) == -1                                                              # kf * cin % 4 (22 channels)
    assert lib.vadx_dfsmn_dft_f(1, C.byref(v(40, 40)), None, None, ptr, C.byref(v(20, 20)), 20, 4, None, None) == -1
    assert b"missing lo" in lib.vadx_last_error()
    two = (C.c_void_p * 2)(ptr, ptr)
    assert lib.vadx_dfsmn_lstm_t(0, C.byref(v(20, 20)), C.byref(ln), C.byref(two), C.byref(two), C.byref(two), C.byref(two), ptr, ptr,
                                 C.byref(v(20, 20)), C.byref(v(20, 20)), 150, 101, 1, None) == -1
    assert b"multiple of 16" in lib.vadx_last_error()
    # the arithmetic rides with the call: F16X2 without the two range-flag words, or an arithmetic an entry point has no kernel for, is
    # refused on the host
    lt = lambda ar, flag: lib.vadx_dfsmn_lstm_t_ex(0, C.byref(v(20, 20)), C.byref(ln), C.byref(two), C.byref(two), C.byref(two), C.byref(two),   # noqa: E731
                                                   ptr, ptr, C.byref(v(20, 20)), C.byref(v(20, 20)), 160, 101, 1, 104, None, ar, flag)
    assert lt(_lib.ARITH["h2"], None) == -1 and b"range flag" in lib.vadx_last_error()
    assert lt(_lib.ARITH["split"], ptr) == -1
    lf = lambda ar, flag: lib.vadx_dfsmn_lstm_f(C.byref(v(40, 40)), C.byref(ln), C.byref(two), C.byref(two), C.byref(two), C.byref(two),         # noqa: E731
                                                C.byref(v(40, 40)), 81, 4, None, ar, flag)
    assert lf(_lib.ARITH["h2"], None) == -1 and b"range flag" in lib.vadx_last_error()
    mc = _lib.MarbleNetCfg(_lib.ARITH["h2"], 0, None)
    blk = lambda cfgp: lib.vadx_marblenet_block2(64, 15, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, 1, 32, None, cfgp)                 # noqa: E731
    assert blk(C.byref(mc)) == -1 and b"range_flag" in lib.vadx_last_error()
    mc.arithmetic = _lib.ARITH["split"]
    assert blk(C.byref(mc)) == -1 and b"arithmetic" in lib.vadx_last_error()
    assert lib.vadx_marblenet_tail(ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, 1, 32, None, C.byref(mc)) == -1
    assert lib.vadx_frag_h2_host(ptr, 16, 32, 7, ptr, None) == -1 and b"k_order" in lib.vadx_last_error()


def test_weight_validation():
    w = weights.silero_synthetic(1)
    assert weights.silero_check(w)
    bad = dict(w)
    bad["enc1_w"] = bad["enc1_w"][:, :, :2]
    with pytest.raises(ValueError):
        weights.silero_check(bad)
    del bad["enc1_w"]
    with pytest.raises(ValueError):
        weights.silero_check(bad)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vadx import silero
    with pytest.raises(_lib.VadxError):
        silero.SileroEngine(None)


def test_host_timestamp_helpers_match_golden(golden):
    g = golden("host")
    for i in range(int(g["n_cases"])):
        fd = [0.01, 0.02][i % 2]
        raw = timestamps.vad_to_timestamps(list(g[f"flags_{i}"]), fd)
        assert np.array_equal(np.array(raw, dtype=np.float64).reshape(-1, 2), g[f"raw_{i}"])
        proc = timestamps.process_timestamps(raw, 0.3, [0.2, 0.25][i % 2])
        assert np.array_equal(np.array(proc, dtype=np.float64).reshape(-1, 2), g[f"proc_{i}"])
    for v, s in zip(g["fmt_in"], g["fmt_out"]):
        assert timestamps.format_time(float(v)) == str(s)
    assert np.array_equal(timestamps.normalize_to_int16(g["norm_in"]), g["norm_out"])
    t = golden("dfsmn_golden_txt")
    assert timestamps.indices([(5.42, 294 * 0.02 + 0.02)], 16000)[0][1] == 94399
    assert timestamps.format_time(2.28) == str(t["seconds"][0]).split(" --> ")[0]


def test_wav_ingest_matches_reference_sample_length():
    """vad_sample.wav (48 kHz stereo s16, 268,292 frames) -> 89,431 mono samples at 16 kHz through
    audioop.tomono + ratecv, the path pydub takes in every reference driver."""
    from vadx import audio_io
    a = audio_io.load_wav(os.path.join(ROOT, "tests", "golden", "vad_sample.wav"))
    assert a.dtype == np.int16 and a.shape == (89431,)
    assert int(np.abs(a).max()) == 4220


def test_firered_checkpoint_loader_matches_reference(golden, tmp_path):
    """model.pth.tar + cmvn.ark -> engine weight dict, against what the reference's load_cmvn + DetectModel.from_pretrained
    produce from the same files (tests/golden/make_golden.py: gen_firered_ckpt).  Pure host code."""
    import types

    import torch

    import _containers
    from vadx import checkpoints as ck
    g = golden("firered_ckpt")
    cfg = dict(weights.FIRERED_CFG, R=3, M=2, H=64, P=32, N1=8, S1=1, N2=4, S2=1, odim=3)
    w = weights.firered_synthetic(21, cfg)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))      # noqa: E731
    sd = {"dfsmn.fc1.0.weight": w["fc1_w"], "dfsmn.fc1.0.bias": w["fc1_b"], "dfsmn.fc2.0.weight": w["fc2_w"],
          "dfsmn.fc2.0.bias": w["fc2_b"], "dfsmn.fsmn1.lookback_filter.weight": w["fsmn0_lb"][:, None, :],
          "dfsmn.fsmn1.lookahead_filter.weight": w["fsmn0_la"][:, None, :], "out.weight": w["out_w"], "out.bias": w["out_b"]}
    for r in range(1, cfg["R"]):
        p = f"dfsmn.fsmns.{r - 1}."
        sd[p + "fc1.0.weight"], sd[p + "fc1.0.bias"], sd[p + "fc2.weight"] = w[f"blk{r}_fc1_w"], w[f"blk{r}_fc1_b"], w[f"blk{r}_fc2_w"]
        sd[p + "fsmn.lookback_filter.weight"] = w[f"fsmn{r}_lb"][:, None, :]
        sd[p + "fsmn.lookahead_filter.weight"] = w[f"fsmn{r}_la"][:, None, :]
    for m in range(cfg["M"]):
        sd[f"dfsmn.dnns.{2 * m}.weight"], sd[f"dfsmn.dnns.{2 * m}.bias"] = w[f"dnn{m}_w"], w[f"dnn{m}_b"]
    torch.save({"args": types.SimpleNamespace(**cfg), "model_state_dict": {k: T(v) for k, v in sd.items()}}, tmp_path / "model.pth.tar")
    for binary in (True, False):
        _containers.write_kaldi_matrix(str(tmp_path / "cmvn.ark"), g["stats"], binary=binary)
        means, inv_std = ck.load_cmvn(str(tmp_path / "cmvn.ark"))
        assert np.array_equal(means, g["means"]) and np.array_equal(inv_std, g["inv_std"])
        got = ck.load_firered(str(tmp_path))
        assert got["cfg"] == cfg
        for k in ("fc1_w", "fc1_b", "blk1_fc2_w", "fsmn2_la", "dnn1_w", "out_w"):
            assert np.array_equal(got[k], g[k]), k
    # a streaming checkpoint (no look-ahead filters) yields N2 = 0
    sd_s = {k: T(v) for k, v in sd.items() if "lookahead" not in k}
    got = ck.firered_from_state(types.SimpleNamespace(**cfg), sd_s)
    assert got["cfg"]["N2"] == 0 and "fsmn0_la" not in got and np.array_equal(got["fc1_w"], w["fc1_w"])


def build_c_client(out_dir):
    """gcc (C99, no hipcc) builds tests/c/cabi_silero.c against include/vadx.h and libvadx.so; returns the executable."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(out_dir), "cabi_silero")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c", "cabi_silero.c"), "-o", exe,
                           "-L" + libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/vadx.h must compile as C99 and every symbol the C client uses must resolve in libvadx.so (the run itself
    needs a GPU: tests/test_gpu_cabi_c.py)."""
    assert os.path.exists(build_c_client(tmp_path))



def test_no_cross_swizzled_packed_f32_in_product_kernels():
    """DESIGN 4e / VERDICT r5 weak 4: a compiler-formed `v_pk_add_f32` whose second source is read with the CROSS swizzle
    (op_sel:[0,1] op_sel_hi:[1,0]) gives wrong sums inside silero_encode_h2_kernel at four waves per SIMD (still reproducible: build with
    -DH2_PK_NATURAL=1, tests/probes/pk_hazard.py), while the same instruction -- own destination, in place, fresh LDS data behind partial
    waits, sources overwritten at once -- is exact in the standalone reproducer (tests/hip/pk_hazard.hip): the root cause is not
    established, so the pattern is BANNED from the product.  The library is built with -fno-slp-vectorize (the SLP vectoriser is what forms
    the swizzled forms; vadx.build.FLAGS) and this test disassembles every gfx950 code object in libvadx.so (tools/pk_scan.py) and fails on
    any packed float32 op with a cross-swizzled VGPR source.  Broadcast swizzles (one half feeding both results: scalar x vector) are
    what every compiler emits and stay."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pk_scan", os.path.join(root, "tools", "pk_scan.py"))
    pk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pk)
    res = pk.scan(_lib.LIB_PATH)
    assert len(res) >= 60 and sum(c["pk"] for c in res.values()) > 1000          # the scan saw the kernels and their packed ops
    bad = {k: c["cross_lines"][:3] for k, c in res.items() if c["cross"]}
    assert not bad, bad
    # the scanner itself: the test-hook reproducer DOES carry the pattern (forced in inline asm), and the scan must find it
    from vadx import build as vbuild
    hooks = pk.scan(vbuild.build_test_hooks(verbose=False))
    assert sum(c["cross"] for c in hooks.values()) >= 16


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: the product package, the shim and tools/ must not import it; bench.py / bench_models.py may,
    inside their cpu_baseline*() functions only; __graft_entry__ in build() / smoke() only."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(_lib.LIB_PATH)

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for fn in ast.walk(tree):
            scope = fn.name if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)) else None
            if scope is None and not isinstance(fn, ast.Module):
                continue
            for node in (fn.body if scope else [n for n in tree.body if not isinstance(n, (ast.FunctionDef, ast.ClassDef))]):
                for sub in ast.walk(node):
                    names = ([a.name for a in sub.names] if isinstance(sub, ast.Import) else
                             [sub.module or ""] if isinstance(sub, ast.ImportFrom) else [])
                    if any(n == "oracle" or n.startswith("oracle.") for n in names):
                        hits.append(scope or "<module>")
        return set(hits)

    product = [os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith(".py")] + [os.path.join(root, "vadx.py")]
    product += [os.path.join(root, "tools", f) for f in os.listdir(os.path.join(root, "tools")) if f.endswith(".py")]      # the development tools too
    for path in product:
        assert not oracle_imports(path), path
    assert oracle_imports(os.path.join(root, "bench.py")) <= {"cpu_baseline"}
    hits = oracle_imports(os.path.join(root, "bench_models.py"))
    assert hits and all(h.startswith("cpu_baseline_") for h in hits), hits
    assert oracle_imports(os.path.join(root, "__graft_entry__.py")) <= {"build", "smoke"}


def test_audio_io_numpy_fallback_matches_audioop(tmp_path):
    """audio_io without the stdlib audioop (removed in Python 3.13): the numpy tomono / ratecv / lin2lin give the same bytes."""
    import audioop
    import wave
    from vadx import audio_io
    rng = np.random.default_rng(5)
    for nch, width, rate, n in ((2, 2, 48000, 7001), (1, 2, 44100, 5000), (2, 2, 16000, 999), (1, 2, 8000, 4000), (1, 1, 22050, 3000),
                                (2, 4, 32000, 2001), (1, 3, 48000, 1500)):
        pcm = rng.integers(-2 ** (8 * width - 1), 2 ** (8 * width - 1) - 1, n * nch)
        raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in pcm)
        path = str(tmp_path / f"a_{nch}_{width}_{rate}.wav")
        with wave.open(path, "wb") as w:
            w.setnchannels(nch); w.setsampwidth(width); w.setframerate(rate); w.writeframes(raw)
        want = audio_io.load_wav(path, 16000)
        saved, audio_io._audioop = audio_io._audioop, None
        try:
            got = audio_io.load_wav(path, 16000)
        finally:
            audio_io._audioop = saved
        assert saved is audioop and got.dtype == np.int16 and np.array_equal(got, want), (nch, width, rate)


@pytest.mark.parametrize("preset,kind", [("fsmn", 2), ("marblenet", 1), ("firered", 1)])
def test_folded_dft_tables_reproduce_the_reference_table(lib, preset, kind):
    """Table-level proof of the folded DFT product on the host: the blob's symmetric tables E / O (over mirror pairs of taps) and
    its f16 residual RX / IX give, in double arithmetic on random frames, the power spectrum of the reference's own float32
    table to ~1e-7 -- i.e. fold + residual is that table, up to the f16 rounding of a term that is itself ~6e-5 of the sum."""
    from vadx import frontend, tables
    p = frontend.PRESETS[preset]
    n_fft, win, hop = p["n_fft"], p["win"], p["hop"]
    cos_t, sin_t = tables.windowed_dft(n_fft, tables.analysis_window(p["window"], win, n_fft, p["variant"]), p["variant"])
    c, s = tables.as_np(cos_t), tables.as_np(sin_t)
    cfg = _lib.FrontendCfg()
    cfg.prep, cfg.k0, cfg.k1 = p["prep"], p["k"][0], p["k"][1]
    cfg.center_pad = n_fft // 2 if p["center"] else 0
    cfg.tap0, cfg.taps = ((n_fft - win) // 2 if win < n_fft else 0), min(win, n_fft)
    cfg.hop, cfg.n_bins, cfg.n_mels, cfg.log_mode, cfg.log_floor = hop, n_fft // 2 + 1, 80, p["log_mode"], p["log_floor"]
    cfg.frames, cfg.window_len = 101, 16000
    assert lib.vadx_frontend_fold_kind(ctypes.byref(cfg), c.ctypes.data, s.ctypes.data, n_fft) == kind
    dense_floats = lib.vadx_frontend_packed_floats(ctypes.byref(cfg))
    cfg.fold = kind
    n = lib.vadx_frontend_packed_floats(ctypes.byref(cfg))
    blob, mel_kb = np.zeros(n, np.float32), np.zeros(10, np.int32)
    fb = np.zeros((80, cfg.n_bins), np.float32)
    assert lib.vadx_frontend_pack_host(ctypes.byref(cfg), c.ctypes.data, s.ctypes.data, n_fft, fb.ctypes.data, blob.ctypes.data,
                                       mel_kb.ctypes.data) == 0
    XLD, nyq = 68, int(cfg.n_bins % 16 == 1)
    nbt = cfg.n_bins // 16 if nyq else (cfg.n_bins + 15) // 16
    plan = blob[n - 52:n - 20].view(np.int32).reshape(8, 4)                    # the blob ends with [8 regions][4] + [8 mel tiles][2] + [4] int32
    # a last-bin (Nyquist) tile keeps one part: the even one about an integer centre, the odd one about a half-integer centre
    assert int(blob[n - 4:].view(np.int32)[0]) == ((2 if kind == 2 else 1) if cfg.n_bins % 16 == 1 else 0)
    regions = [r for r in plan if r[0] > 0]
    Pb, Kb32 = int(sum(r[0] for r in regions)), (cfg.taps + 31) // 32
    ka, kb = [], []                                   # tap of either member of pair p; -1 = the row of zeros
    for blocks, offA, offB, strB in regions:
        for t in range(16 * blocks):
            ra, rb = offA // XLD + t, offB // XLD + (t if strB > 0 else -t)
            ka.append((offA % XLD) * hop + ra if ra < hop else -1)
            kb.append((offB % XLD) * hop + rb if rb < hop else -1)
    ka, kb = np.array(ka), np.array(kb)
    fold = blob[dense_floats:dense_floats + (nbt + nyq) * 32 * Pb * 16]
    r_i, k_i = np.meshgrid(np.arange((nbt + nyq) * 32), np.arange(Pb * 16), indexing="ij")
    ldw = Pb * 16
    fm = fold[(r_i // 16) * 16 * ldw + (k_i // 16) * 256 + (((k_i % 16) // 4) * 16 + r_i % 16) * 4 + k_i % 4].astype(np.float64)
    res = blob[dense_floats + fold.size:n - 52].view(np.float16).astype(np.float64).reshape(nbt + nyq, 2, Kb32, 4, 16, 8) / 8192.0
    res = res.transpose(0, 1, 4, 2, 3, 5).reshape(nbt + nyq, 2, 16, Kb32 * 32)[..., :cfg.taps]        # [tile][RX|IX][bin in tile][tap]
    rng = np.random.default_rng(kind)
    x = rng.standard_normal((5, cfg.taps))
    xz = np.concatenate([x, np.zeros((5, 1))], axis=1)                       # index -1 = 0
    valid = (ka >= 0)[None, :]
    u, v = np.where(valid, xz[:, ka] + xz[:, kb], 0.0), np.where(valid, xz[:, ka] - xz[:, kb], 0.0)
    R, I = c[:, cfg.tap0:cfg.tap0 + cfg.taps].astype(np.float64), s[:, cfg.tap0:cfg.tap0 + cfg.taps].astype(np.float64)
    want = (x @ R.T) ** 2 + (x @ I.T) ** 2                                    # [frame][bin]
    got = np.zeros_like(want)
    for t in range(nbt):
        bins = np.arange(16 * t, min(16 * t + 16, cfg.n_bins - nyq))
        re = u @ fm[t * 32:t * 32 + 16].T + x @ res[t, 0].T
        im = v @ fm[t * 32 + 16:t * 32 + 32].T + x @ res[t, 1].T
        got[:, bins] = (re ** 2 + im ** 2)[:, :len(bins)]
    if nyq:                                           # the last bin's own tile: u x row 0 -> re', v x row 17 -> im'; residual rows 0, 1 of part 0
        re = u @ fm[nbt * 32] + x @ res[nbt, 0, 0]
        im = v @ fm[nbt * 32 + 17] + x @ res[nbt, 0, 1]
        got[:, -1] = re ** 2 + im ** 2
    scale = (np.abs(x) @ np.abs(R).T) ** 2
    assert np.abs(got - want).max() / scale.max() < 2e-7
    assert (np.abs(got - want) / scale).max() < 2e-7


def test_kind3_fold_tables_reproduce_the_reference_table(lib):
    """FSMN's periodic window admits the opt-in time x frequency fold (kind 3: four sums per bin b < 128 give bins b and 256 - b).  Same
    table-level proof as above: the blob's E / O over the kernel's own slot plan plus the four f16 residual tables per row tile, in
    double on random frames, give the power spectrum of the reference's float32 table for all 257 bins."""
    from vadx import frontend, tables
    p = frontend.PRESETS["fsmn"]
    n_fft, win, hop = p["n_fft"], p["win"], p["hop"]
    cos_t, sin_t = tables.windowed_dft(n_fft, tables.analysis_window(p["window"], win, n_fft, p["variant"]), p["variant"])
    c, s = tables.as_np(cos_t), tables.as_np(sin_t)
    cfg = _lib.FrontendCfg()
    cfg.prep, cfg.k0, cfg.k1 = p["prep"], p["k"][0], p["k"][1]
    cfg.center_pad, cfg.tap0, cfg.taps = n_fft // 2, (n_fft - win) // 2, win
    cfg.hop, cfg.n_bins, cfg.n_mels, cfg.log_mode, cfg.log_floor = hop, n_fft // 2 + 1, 80, p["log_mode"], p["log_floor"]
    cfg.frames, cfg.window_len = 101, 16000
    assert lib.vadx_frontend_fold_kind(ctypes.byref(cfg), c.ctypes.data, s.ctypes.data, n_fft) == 2      # kind 3 is opt-in, never the default answer
    dense_floats = lib.vadx_frontend_packed_floats(ctypes.byref(cfg))
    cfg.fold = 3
    n = lib.vadx_frontend_packed_floats(ctypes.byref(cfg))
    blob, mel_kb = np.zeros(n, np.float32), np.zeros(10, np.int32)
    fb = np.zeros((80, cfg.n_bins), np.float32)
    assert lib.vadx_frontend_pack_host(ctypes.byref(cfg), c.ctypes.data, s.ctypes.data, n_fft, fb.ctypes.data, blob.ctypes.data,
                                       mel_kb.ctypes.data) == 0
    XLD, taps, nq, ntl, Kb32 = 68, cfg.taps, n_fft // 4, n_fft // 64 + 1, (cfg.taps + 31) // 32
    # sizes: fold matrix ntl*32*Pb*16 floats, residual ntl*4*Kb32*256 floats, plan Pb*32 ints, bands 16 ints
    Pb = (n - dense_floats - 16 - ntl * 4 * Kb32 * 256) // (ntl * 32 * 16 + 32)
    assert dense_floats + ntl * 32 * Pb * 16 + ntl * 4 * Kb32 * 256 + Pb * 32 + 16 == n and Pb % 2 == 0
    o_fold, o_res = dense_floats, dense_floats + ntl * 32 * Pb * 16
    o_plan = o_res + ntl * 4 * Kb32 * 256
    plan = blob[o_plan:o_plan + Pb * 32].view(np.int32).reshape(Pb, 4, 2, 4)          # [block][q][A|B][j]
    def tap(off):
        r, col = off // XLD, off % XLD
        return np.where(r < hop, col * hop + r, -1)
    ka = tap(plan[:, :, 0, :]).reshape(Pb * 16)                                         # contraction index 16 S + 4 q + j
    kb = tap(plan[:, :, 1, :]).reshape(Pb * 16)
    ldw = Pb * 16
    r_i, k_i = np.meshgrid(np.arange(ntl * 32), np.arange(ldw), indexing="ij")
    fm = blob[o_fold:o_res][(r_i // 16) * 16 * ldw + (k_i // 16) * 256 + (((k_i % 16) // 4) * 16 + r_i % 16) * 4 + k_i % 4].astype(np.float64)
    res = blob[o_res:o_plan].view(np.float16).astype(np.float64).reshape(ntl, 4, Kb32, 4, 16, 8) / 8192.0
    res = res.transpose(0, 1, 4, 2, 3, 5).reshape(ntl, 4, 16, Kb32 * 32)[..., :taps]  # [tile][accumulator][bin in tile][tap]
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5, taps))
    xz = np.concatenate([x, np.zeros((5, 1))], axis=1)
    u, v = xz[:, ka] + xz[:, kb], xz[:, ka] - xz[:, kb]
    even = np.arange(ldw) < ldw // 2
    R, I = c[:, cfg.tap0:cfg.tap0 + taps].astype(np.float64), s[:, cfg.tap0:cfg.tap0 + taps].astype(np.float64)
    want = (x @ R.T) ** 2 + (x @ I.T) ** 2
    got = np.zeros_like(want)
    for t in range(ntl - 1):
        E, O = fm[t * 32:t * 32 + 16], fm[t * 32 + 16:t * 32 + 32]
        A = (u * even) @ E.T + x @ res[t, 0].T
        C = (v * even) @ O.T + x @ res[t, 1].T
        B = (u * ~even) @ E.T + x @ res[t, 2].T
        D = (v * ~even) @ O.T + x @ res[t, 3].T
        bins = np.arange(16 * t, 16 * t + 16)
        got[:, bins] = (A + B) ** 2 + (C + D) ** 2
        got[:, 2 * nq - bins] = (A - B) ** 2 + (C - D) ** 2
    t = ntl - 1
    re = u @ fm[t * 32] + x @ res[t, 0, 0]
    im = v @ fm[t * 32 + 17] + x @ res[t, 0, 1]
    got[:, nq] = re ** 2 + im ** 2
    scale = (np.abs(x) @ np.abs(R).T) ** 2
    assert (np.abs(got - want) / scale).max() < 2e-7, (np.abs(got - want) / scale).max()


def test_kind3_fold_is_refused_where_the_table_does_not_admit_it(lib):
    from vadx import frontend, tables
    p = frontend.PRESETS["marblenet"]                       # symmetric Hann: mirror pairs mix the parities
    cos_t, sin_t = tables.windowed_dft(512, tables.analysis_window(p["window"], 400, 512, p["variant"]), p["variant"])
    c, s = tables.as_np(cos_t), tables.as_np(sin_t)
    cfg = _lib.FrontendCfg()
    cfg.prep, cfg.k0, cfg.k1 = p["prep"], p["k"][0], p["k"][1]
    cfg.center_pad, cfg.tap0, cfg.taps = 256, 56, 400
    cfg.hop, cfg.n_bins, cfg.n_mels, cfg.log_mode, cfg.log_floor = 160, 257, 80, p["log_mode"], p["log_floor"]
    cfg.frames, cfg.window_len, cfg.fold = 101, 16000, 3
    n = lib.vadx_frontend_packed_floats(ctypes.byref(cfg))
    assert n > 0                                            # the geometry has a plan; the TABLE is what fails
    blob, mel_kb = np.zeros(n, np.float32), np.zeros(10, np.int32)
    fb = np.zeros((80, 257), np.float32)
    assert lib.vadx_frontend_pack_host(ctypes.byref(cfg), c.ctypes.data, s.ctypes.data, 512, fb.ctypes.data, blob.ctypes.data, mel_kb.ctypes.data) != 0
    assert b"kind-3" in lib.vadx_last_error()
