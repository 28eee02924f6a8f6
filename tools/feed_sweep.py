"""MarbleNet config 4 from pinned host PCM: chunk-size sweep of vadx.feed.HostPcmFeed (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import vadx  # noqa: E402,F401
import bench_models as bm  # noqa: E402
from vadx import feed as vfeed, marblenet, weights  # noqa: E402

dev = torch.device("cuda:0")
eng = marblenet.MarbleNetEngine(weights.marblenet_synthetic(1234), device=dev)
clips, n = 8192, 89431
audio = bm.synth_pcm16(torch, dev, clips, n, seed=1404)
host = vfeed.pin(audio.cpu())
res = bm.device_ms(torch, lambda: eng.run(audio), 3)
print("resident %.2f ms" % res)
a = torch.empty_like(audio)
torch.cuda.synchronize()
t0 = time.perf_counter(); a.copy_(host, non_blocking=True); torch.cuda.synchronize(); up = time.perf_counter() - t0
print("upload alone %.2f ms = %.1f GB/s" % (up * 1e3, clips * n * 2 / up / 1e9))
for chunk in (64, 128, 256, 512, 1024, 2048):
    f = vfeed.HostPcmFeed(dev, n, chunk)
    eng.run_from_host(host, feed=f); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run_from_host(host, feed=f); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("chunk %5d: %.2f ms" % (chunk, sorted(ts)[1] * 1e3), flush=True)
    del f
