#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/prof_dfsmn.sh ; per-kernel time of one DFSMN sub-batch (960 windows x2)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_dfsmn; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 tools/prof_dfsmn.py > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k[-60:]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values())
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print(f"{k:62s} n={len(v):5d} total={sum(v)/1e6:9.2f} ms  mean={sum(v)/len(v)/1e3:9.1f} us  {100*sum(v)/tot:5.1f}%")
    print("total", tot / 1e6, "ms")
PY
find "$OUT" -type f -size +1M -delete
