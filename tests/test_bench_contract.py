"""bench.py's output contract (the driver parses ONE JSON line): keys, units and internal consistency."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_flop_accounting_and_committed_profile():
    sys.path.insert(0, ROOT)
    import bench
    # issued MFMA flops per window: 5504 v_mfma_f32_16x16x4 per 16-window tile, 2048 flop each
    assert bench.FLOP_ENCODE_ISSUED == 5504 * 2048 // 16 == 704512
    assert bench.FLOP_ENCODE == 1186816 and bench.FLOP_RECUR == 131072
    tr = bench.profiled_traffic()                      # newest profiles/r*/SUMMARY.txt, FETCH doubled per the gfx950 note
    assert tr is not None and tr["source"].startswith("profiles/r")
    algorithmic = bench.CLIPS_PER_GPU * bench.STEPS_PER_CLIP * 4096
    assert 0.95 * algorithmic < tr["bytes"] < 1.10 * algorithmic      # no wasted re-reads


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--clips", "64", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert REQUIRED <= set(d)
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert abs(d["value"] - 64 * 313 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert 0 < ro["frac"] < 1 and "workload" in d["config"]
