"""bench.py's output contract (the driver parses ONE JSON line): keys, units and internal consistency; the N-rank launch
path (self-launch before any GPU call, barrier, max over ranks, rank 0 prints) in a CPU dry run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_flop_and_byte_accounting_and_committed_profile():
    sys.path.insert(0, ROOT)
    import bench
    # issued MFMA flops per window: 4480 v_mfma_f32_16x16x4 per 16-window tile (Winograd conv1), 2048 flop each
    assert bench.FLOP_ENCODE_ISSUED == 4480 * 2048 // 16 == 573440
    assert bench.FLOP_ENCODE == 1186816 and bench.FLOP_RECUR == 131072
    # 2048 B of float32 PCM in + one 4-byte score out per 512-sample window; the gx intermediate is NOT algorithmic
    assert bench.ALGO_BYTES_PER_WINDOW == 2052
    tr = bench.profiled_traffic()                      # newest profiles/r*/SUMMARY.txt, FETCH doubled per the gfx950 note
    assert tr is not None and tr["source"].startswith("profiles/r")
    windows = bench.CLIPS_PER_GPU * bench.STEPS_PER_CLIP
    # the encoder launch moves the PCM in and the gx intermediate out: ~2x the algorithmic bytes, and nothing beyond that
    assert 0.95 * windows * 4096 < tr["bytes"] < 1.10 * windows * 4096
    rec = bench.profiled_traffic("silero_lstm_kernel")
    assert rec is not None and 0.95 * windows * 2048 < rec["bytes"] < 1.10 * windows * 2052
    ratio = (tr["bytes"] + rec["bytes"]) / (windows * bench.ALGO_BYTES_PER_WINDOW)
    assert 2.8 < ratio < 3.3                           # the gx round trip: design traffic, reported as hbm.traffic_ratio
    # the split-product kernel set (round 4, the default) moves the same tensors: its committed profile must say so too
    trs, recs = bench.profiled_traffic("silero_encode_split_kernel"), bench.profiled_traffic("silero_lstm_split_kernel")
    assert trs is not None and recs is not None and trs["source"].startswith(("profiles/r04", "profiles/r05", "profiles/r06"))
    assert 0.95 * windows * 4096 < trs["bytes"] < 1.10 * windows * 4096
    assert 0.95 * windows * 2048 < recs["bytes"] < 1.10 * windows * 2052
    # the fp16 x 2 set (round 5, the default): same tensors; its encoder's first profile (profiles/r05_silero) carried 4.8 GB of scratch
    # traffic from a spilled max chain -- the newest one must not
    trh, rech = bench.profiled_traffic("silero_encode_h2_kernel"), bench.profiled_traffic("silero_lstm_h2_kernel")
    assert trh is not None and rech is not None and trh["source"].startswith("profiles/r06")
    assert 0.95 * windows * 4096 < trh["bytes"] < 1.25 * windows * 4096
    assert 0.95 * windows * 2048 < rech["bytes"] < 1.10 * windows * 2052


def test_secondary_flop_formulas():
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    import bench_models as bm
    assert bm.flop_fsmn_frame() == 2 * 426960                        # 400-140-250-(128 x4, 20-tap FIR)-140-248
    assert bm.flop_marblenet_out_frame() == 2 * 89456                # published 3x2x64 layout, per 20 ms output frame
    assert bm.flop_firered_frame() == 2 * 585984                     # placeholder dims of SURVEY appendix B
    # on fp16 x 2 EVERY 1x1 conv of MarbleNet runs as split products: prologue 10240 + fused blocks 20480 + 12288 + 12288 + tail 8192 + 16384
    # MACs per output frame (the prologue and tail were priced at the f32 peak, which doubled the reported frac)
    assert bm.flop_marblenet_h2_out_frame() == 2 * 79872 < bm.flop_marblenet_out_frame()
    d = bm.flop_dfsmn_window()
    assert d["total"] == sum(v for k, v in d.items() if k not in ("total", "cfb_front", "cfb_back"))      # the fused kernels regroup pw_conv / dft_f work
    assert d["cfb_front"] + d["cfb_back"] < d["pw_conv"] + d["dft_f"] + d["lstm_f"]
    assert 5.5e9 < d["total"] < 6.5e9 and d["lstm_f"] > d["lstm_t"] > d["istft"]
    # per ENTRY POINT: either execution (fused gated blocks or the six-launch chain) accounts for the same ICCRN arithmetic, and the
    # three pw_conv launches the fused pass keeps carry only their own flops (round 3 priced them with all of pw_conv's: frac 2.46)
    net = d["lstm_f"] + d["dft_f"] + d["pw_conv"] + d["lstm_t"]
    for fused in (False, True):
        e = bm.flop_dfsmn_by_entry(fused)
        assert sum(e.values()) == net and all(v >= 0 for v in e.values())
    assert bm.flop_dfsmn_by_entry(True)["pw_conv"] < 0.06 * d["pw_conv"]


def test_cpu_baseline_is_bounded_and_reports_the_protocol():
    """The CPU leg must stay a bounded sample (it once ran for half an hour on a 256-thread host with torch's intra-op pool on
    every thread): a 0.5 s budget finishes within two minutes here and carries the BASELINE.md section-3 protocol figures."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    t0 = time.perf_counter()
    r = bench.cpu_baseline(0.5)
    assert time.perf_counter() - t0 < 120
    assert r["kind"] == "port" and r["value"] > 0 and 1 <= r["cores"] <= 16
    p = r["baseline_md_protocol"]
    assert p["threads"] == r["cores"] and p["model_calls_only"]["value"] > 0 and p["end_to_end"]["rtf"] > 0


def test_self_launch_two_ranks_dry_run():
    """`bench.py --gpus 2` with no launcher starts two ranks itself (torch.distributed.run on 127.0.0.1) and rank 0 prints
    one line with n_gpus == 2; --dry-run swaps RCCL + kernels for gloo + sleep so this runs on a CPU-only box."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1",
                        "--clips", "10"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] is None and d["data"].startswith("dry-run")
    assert d["config"]["shards"] == [[0, 0, 10], [1, 10, 20]]       # contiguous clip shards in rank order
    assert d["c4_sharded"]["shards"] == [[0, 0, 4096], [1, 4096, 8192]] and d["c4_sharded"]["scaling"] == "strong"      # BASELINE config 4
    assert d["ms_per_step"] >= 5.0


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--clips", "64", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary", "--no-c4-sharded", "--detail", os.path.join(ROOT, "gpurun_out", "test_detail.json")],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 6144                                       # the driver records the tail of the line
    d = json.loads(lines[0])
    full = json.load(open(os.path.join(ROOT, "gpurun_out", "test_detail.json")))
    assert full["value"] == d["value"] and "instruction_mix" in full["roofline"]
    # the additive VALU : MFMA ceiling model was measured for f32-input MFMAs: a kernel on split products never carries it
    for key in ("roofline", "roofline_recurrent"):
        mix = full[key].get("instruction_mix")
        if mix is not None and full[key]["arithmetic"] != "f32":
            assert mix["ceiling_frac"] is None and mix["model"].startswith("none"), (key, mix)
        assert full[key]["frac_of_sustained"] >= full[key]["frac"]
    assert REQUIRED <= set(d)
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert abs(d["value"] - 64 * 313 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert 0 < ro["frac"] < 1 and "workload" in d["config"]
    assert ro["algorithmic_bytes_per_launch"] == 64 * 313 * 2052
    assert d["hbm"]["algorithmic_bytes_per_frame"] == 2052 and 0 < d["hbm"]["frac"] < 1
    # the PCIe-inclusive mode ran, and feeding int16 through the in-kernel scaling gave the very same scores
    assert d["feed"]["scores_bit_identical_to_resident_f32_path"] is True and d["feed"]["value"] < d["value"] * 1.05


@pytest.mark.gpu
def test_secondary_workloads_small():
    """The bench_models workloads at toy sizes: every config produces a roofline / hbm entry and a per-entry split."""
    import torch
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    import bench_models as bm
    dev = torch.device("cuda", 0)
    for fn, kw in ((bm.fsmn_c3, dict(clips=32)), (bm.marblenet_c4, dict(clips=32)), (bm.firered_c5, dict(clips=32)),
                   (bm.dfsmn_c5, dict(clips=4, sub_batch=60))):
        out = fn(torch, dev, 1, 0, **kw)
        assert out["ms"] > 0 and out["frames_per_s"] > 0 and 0 < out["roofline"]["frac"] < 1, out
        assert 0 < out["hbm"]["frac"] < 1 and sum(out["kernel_ms"].values()) <= out["ms"] * 1.5
        for key, ro in out.items():                  # no timed kernel may be credited with more work than the pipe can do
            if key.startswith("roofline") and isinstance(ro, dict):
                for r in (ro.values() if key == "roofline_by_entry" else [ro]):
                    assert 0 <= r["frac"] <= 1, (key, r)


@pytest.mark.gpu
def test_marblenet_h2_flops_follow_the_launched_fragments():
    """bench_models prices as fp16 x 2 work exactly the 1x1 convs whose `_h` fragments MarbleNetEngine._run_fused hands to its launches."""
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    import bench_models as bm
    from vadx import marblenet, weights
    eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234))
    assert eng.h2_ok and bm.flop_marblenet_h2_out_frame(eng) == bm.flop_marblenet_h2_out_frame() == 2 * 79872


SECONDARY_TAGS = ("fsmn", "marblenet", "firered", "dfsmn")
# C-ABI entry point (vadx._lib.trace names) -> the kernels it launches on the default arithmetic, as rocprofv3 names them in SUMMARY.txt
ENTRY_KERNELS = {
    "fsmn": {"vadx_frontend_logmel_means": ["frontend_split_kernel<vadx::SchemeH2>"], "vadx_fsmn_clips": ["fsmn_clips_kernel<2>"],
             "vadx_fsmn_window_stats": ["fsmn_stats_kernel"]},
    "marblenet": {"vadx_frontend_logmel": ["frontend_split_kernel<vadx::SchemeH2>"], "vadx_sepconv_block": ["sepconv_block_kernel<11, 1, 2, 2>"],
                  "vadx_marblenet_block2": ["jasper_block2_kernel<13, 2>", "jasper_block2_kernel<15, 3>", "jasper_block2_kernel<17, 3>"],
                  "vadx_marblenet_tail": ["marblenet_tail_kernel<2>"]},
    "firered": {"vadx_frontend_logmel": ["frontend_split_kernel<vadx::SchemeH2>"], "vadx_firered_run": ["firered_kernel<2>"]},
}


def _newest_summary(tag):
    import glob
    import re
    paths = glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}", "SUMMARY.txt"))
    return max(paths, key=lambda p: re.search(r"profiles/r(\d+)([a-z]?)_", p.replace(os.sep, "/")).groups())


@pytest.mark.parametrize("tag", SECONDARY_TAGS)
def test_secondary_traffic_comes_from_the_newest_profile(tag):
    """BENCH_r05's traffic figures of configs 3 - 5 were round 4's, because the round-5 summaries (reduced batches,
    tools/prof_model.sh) had no per-pass lines and the reader silently fell back.  The newest profiles/r*_<tag>/SUMMARY.txt must be a
    BASELINE-size one that the reader accepts, and every kernel the bench looks up by name must be in it."""
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    import bench_models as bm
    newest = os.path.relpath(_newest_summary(tag), ROOT)
    tr = bm.profiled_pass_traffic(tag)
    assert tr is not None and tr["source"] == newest, (tr and tr["source"], newest)
    text = open(os.path.join(ROOT, newest)).read()
    assert "BASELINE-size workload" in text.splitlines()[0]
    for kernels in ENTRY_KERNELS.get(tag, {}).values():
        for k in kernels:
            assert k in text, (tag, k)
            rec = bm.profiled_kernel_traffic(tag, k)
            assert rec is not None and rec["source"] == newest
    if tag == "dfsmn":      # looked up by entry-group name in bench_models.dfsmn_c5
        for k in ("cfb_front", "cfb_back", "lstm_f", "lstm_t", "pw_conv"):
            rec = bm.profiled_kernel_traffic(tag, k)
            assert rec is not None and rec["source"] == newest, k


@pytest.mark.gpu
@pytest.mark.parametrize("tag", [t for t in SECONDARY_TAGS if t in ENTRY_KERNELS])
def test_profiled_kernels_are_the_ones_the_bench_pass_launches(tag):
    """The entry points a bench pass goes through (vadx._lib.trace) are exactly those whose kernels the newest committed summary lists:
    a renamed or replaced kernel makes the committed traffic figures stale, and this test red."""
    import torch
    sys.path.insert(0, ROOT)
    import vadx  # noqa: F401
    import bench_models as bm
    fn = {"fsmn": bm.fsmn_c3, "marblenet": bm.marblenet_c4, "firered": bm.firered_c5}[tag]
    out = fn(torch, torch.device("cuda", 0), 1, 0, clips=32)
    assert set(out["kernel_ms"]) == set(ENTRY_KERNELS[tag]), (sorted(out["kernel_ms"]), sorted(ENTRY_KERNELS[tag]))
    text = open(_newest_summary(tag)).read()
    for entry, kernels in ENTRY_KERNELS[tag].items():
        for k in kernels:
            assert k in text, (entry, k)
    for key in ("roofline", "roofline_net"):
        if key in out and out[key].get("traffic_source"):
            assert out[key]["traffic_source"] == os.path.relpath(_newest_summary(tag), ROOT)
