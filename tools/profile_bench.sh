#!/bin/bash
# Profile bench.py under rocprofv3 (run on the GPU box via gpurun): kernel-trace stats + separate PMC passes.
# usage: bash tools/profile_bench.sh <tag>
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
run() {  # name, steps, warmup, extra rocprof args...
  local name=$1 steps=$2 warm=$3; shift 3
  timeout 600 rocprofv3 --kernel-trace --output-format csv "$@" -d "$OUT/$name" -o "$name" -- \
      python3 bench.py --steps "$steps" --warmup "$warm" --no-cpu-baseline --no-secondary --no-feed --no-c4-sharded --detail "" > "$OUT/$name.bench.json" 2> "$OUT/$name.err.log"
  echo "== $name rc=$?"; tail -c 300 "$OUT/$name.bench.json"; echo
}
run stats 5 2 --stats
run fetch 2 1 --pmc FETCH_SIZE
run write 2 1 --pmc WRITE_SIZE
run sq    2 1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE
run sq2   2 1 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run mem   2 1 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
csv.field_size_limit(1 << 30)
out = sys.argv[1]
summ = open(os.path.join(out, "SUMMARY.txt"), "w")
def p(*a):
    s = " ".join(str(x) for x in a); print(s); summ.write(s + "\n")
def short(k):
    k = k.split("(")[0]
    return k[-44:]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    p("# rocprofv3 --kernel-trace --stats :", os.path.relpath(f, out))
    for row in csv.reader(open(f)):
        if any("vadx" in c or c == "Name" for c in row): p(",".join(row))
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "vadx" not in row["Kernel_Name"]: continue
        acc[(short(row["Kernel_Name"]), int(row["Grid_Size_X"]))].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    p("# per (kernel, grid) dispatch durations from the kernel trace [ns]: n, mean, median, min, max")
    for (k, g), v in sorted(acc.items()): p(f"{k:46s} grid={g:10d} n={len(v):3d} mean={sum(v)/len(v):14.1f} median={sorted(v)[len(v)//2]:12d} min={min(v):12d} max={max(v):12d}")
for name in ("fetch", "write", "sq", "sq2", "mem"):
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "vadx" not in k: continue
            key = (short(k), int(row["Grid_Size"]), row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
        p("# pmc pass", name, "(mean per dispatch, grouped by grid size)")
        for (k, g, c), (v, n) in sorted(acc.items()): p(f"{k:46s} grid={g:10d} {c:26s} {v / n:18.1f}  n={n}")
summ.close()
PY
# keep only small text artefacts
find "$OUT" -type f -size +2M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
