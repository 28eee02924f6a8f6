"""DFSMN config-5 pass at full size, per-entry split; VADX_LIBRARY selects the build (A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx  # noqa: F401
import bench_models as bm
r = bm.dfsmn_c5(torch, torch.device("cuda", 0), 2, 0)
print("DFSMN", os.path.basename(os.environ.get("VADX_LIBRARY", "libvadx.so")), "ms %.1f" % r["ms"], {k: round(v, 1) for k, v in r["kernel_ms"].items() if v > 30})
