"""FSMN / FireRed / MarbleNet net kernels at reduced batch, per-entry split; VADX_LIBRARY selects the build (A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vadx  # noqa: F401
import bench_models as bm
dev = torch.device("cuda", 0)
f = bm.fsmn_c3(torch, dev, 3, 0, clips=1024)
r = bm.firered_c5(torch, dev, 3, 0, clips=512)
m = bm.marblenet_c4(torch, dev, 3, 0, clips=2048)
print("NETS", os.path.basename(os.environ.get("VADX_LIBRARY", "libvadx.so")), "fsmn %.2f" % f["kernel_ms"]["vadx_fsmn_clips"],
      "firered %.2f" % r["kernel_ms"]["vadx_firered_run"], "marblenet", {k: round(v, 2) for k, v in m["kernel_ms"].items() if "frontend" not in k})
