#!/bin/bash
# A/B of the five BASELINE configs between the in-tree libvadx.so and a library built with -fno-slp-vectorize (gpurun box):
#   bash tools/ab_noslp.sh <other.so>    -> gpurun_out/ab_noslp_{a,b}.json (+ one-line summaries)
cd "$GRAFT_REPO_ROOT"
other=$1
for tag in a b a b; do
  lib=""; [ $tag = b ] && lib=$other
  VADX_LIBRARY=$lib python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-feed --no-c4-sharded --detail gpurun_out/ab_noslp_$tag.json > gpurun_out/ab_noslp_$tag.line 2> gpurun_out/ab_noslp_$tag.err
  python3 - gpurun_out/ab_noslp_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); s = d.get("secondary", {})
print(sys.argv[2], "C2 ms/step %.3f" % d["ms_per_step"], " ".join(f"{k} {v['ms']:.2f}" for k, v in s.items() if isinstance(v, dict) and "ms" in v), flush=True)
PY
done
