#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/prof_model.sh <model> ; prints per-kernel time + SQ counters
M=${1:-firered}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$M; rm -rf "$OUT"; mkdir -p "$OUT"
PASSES=${PASSES:-"stats sq sq2 mem fetch write"}      # FETCH_SIZE and WRITE_SIZE need a pass each (together they return nothing)
pmc() { timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -- python3 tools/prof_model.py "$M" 1 > /dev/null 2>&1; }
for p in $PASSES; do
  case $p in
    stats) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 tools/prof_model.py "$M" 2 > /dev/null 2>&1 ;;
    sq)    pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE -d "$OUT/sq" -o s ;;
    sq2)   pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d "$OUT/sq2" -o s ;;
    mem)   pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT/mem" -o s ;;
    fetch) pmc FETCH_SIZE -d "$OUT/fetch" -o s ;;
    write) pmc WRITE_SIZE -d "$OUT/write" -o s ;;
  esac
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:8]:
        if "vadx" in r["Name"]: print(r["Name"].split("(")[0][-50:].ljust(52), r["Calls"].rjust(4), "%9.3f ms avg" % (float(r["AverageNs"]) / 1e6), r["Percentage"] + "%")
for name in ("sq", "sq2", "mem", "fetch", "write"):
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "vadx" not in row["Kernel_Name"]: continue
            k = (row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (k, c), (v, n) in sorted(acc.items()): print(f"{k:42s} {c:26s} {v / n:16.0f} n={n}")
PY
find "$OUT" -type f -size +1M -delete
