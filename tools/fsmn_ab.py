"""FSMN config-3 pass at 1024 clips, per-entry split; VADX_LIBRARY selects the build (A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx  # noqa: F401
import bench_models as bm
r = bm.fsmn_c3(torch, torch.device("cuda", 0), 3, 0, clips=1024)
print("FSMN", os.path.basename(os.environ.get("VADX_LIBRARY", "libvadx.so")), "ms %.2f" % r["ms"], {k: round(v, 2) for k, v in r["kernel_ms"].items()})
