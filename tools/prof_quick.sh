#!/bin/bash
# quick traffic check of one secondary workload: kernel times + FETCH_SIZE / WRITE_SIZE (two rocprofv3 passes)   usage: prof_quick.sh <model> <tag>
M=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pq_${TAG}_$M; rm -rf "$OUT"; mkdir -p "$OUT"
# one counter per pass: FETCH_SIZE + WRITE_SIZE together exceed what one pass can collect (rocprofv3 aborts and then hangs)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/fetchwrite/$c" -o "$c" -- python3 tools/prof_secondary.py "$M" 2 > "$OUT/$c.out" 2> "$OUT/$c.err.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
out = sys.argv[1]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/fetchwrite/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-48:]
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:50s} {c:12s} per pass {v / 2 / 1048576:10.2f} GiB-units(KiB/2^20)  dispatches/pass {n[(k, c)] // 2}")
tr = collections.defaultdict(list)
for f in glob.glob(out + "/fetchwrite/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        tr[row["Kernel_Name"].split("(")[0][-48:]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for k, v in sorted(tr.items(), key=lambda kv: -sum(kv[1])):
    if "vadx" in k or "fsmn" in k: print(f"{k:50s} total ms per pass {sum(v) / 2 / 1e6:9.3f}")
PY
find "$OUT" -type f -size +2M -delete
