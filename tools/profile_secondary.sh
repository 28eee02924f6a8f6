#!/bin/bash
# Profile the secondary BASELINE configs under rocprofv3 (run on the GPU box via gpurun): kernel-trace stats + separate PMC passes.
# usage: bash tools/profile_secondary.sh <tag> [models...]      -> gpurun_out/prof_<tag>_<model>/SUMMARY.txt
set -u
TAG=${1:-r02}; shift
MODELS=${*:-"fsmn marblenet firered dfsmn"}
export PASSES=${PASSES:-3}      # passes per rocprofv3 run, after the clock ramp of prof_secondary.py
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for M in $MODELS; do
  OUT=gpurun_out/prof_${TAG}_$M
  rm -rf "$OUT"; mkdir -p "$OUT"
  run() {  # name, extra rocprof args...
    local name=$1; shift
    timeout 900 rocprofv3 --kernel-trace --output-format csv "$@" -d "$OUT/$name" -o "$name" -- \
        python3 tools/prof_secondary.py "$M" "$PASSES" > "$OUT/$name.out" 2> "$OUT/$name.err.log"
    echo "== $M $name rc=$? $(tail -1 "$OUT/$name.out")"
  }
  run stats --stats
  run fetch --pmc FETCH_SIZE
  run write --pmc WRITE_SIZE
  run sq    --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE
  run sq2   --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
  python3 - "$OUT" "$M" <<'PY'
import csv, glob, os, sys, collections
csv.field_size_limit(1 << 30)
out, model = sys.argv[1], sys.argv[2]
PASSES = int(os.environ.get('PASSES', '3'))
summ = open(os.path.join(out, "SUMMARY.txt"), "w")
def p(*a):
    s = " ".join(str(x) for x in a); print(s); summ.write(s + "\n")
def short(k):
    k = k.split("(")[0]
    return k.replace("void ", "")[-64:]
p(f"# {model}: BASELINE-size workload of bench_models.py, {PASSES} passes per rocprofv3 run (tools/prof_secondary.py); every number below is PER PASS")
p("# (sum over all dispatches of the kernel, divided by the number of passes; FETCH_SIZE / WRITE_SIZE are KiB, FETCH not yet doubled)")
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "vadx" not in row["Kernel_Name"]: continue
        acc[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values())
    p("# rocprofv3 --kernel-trace --stats: per kernel, per pass: dispatches, total ms, mean us, share")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        p(f"{k:66s} calls_per_pass={len(v) // PASSES:5d} total_ms={sum(v) / PASSES / 1e6:10.3f} mean_us={sum(v) / len(v) / 1e3:10.1f} share={100 * sum(v) / tot:5.1f}%")
    p(f"# all vadx kernels: {tot / PASSES / 1e6:.3f} ms per pass")
for name in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "vadx" not in k: continue
            key = (short(k), row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
        p("# pmc pass", name, "(sum over the kernel's dispatches, per pass)")
        for (k, c), (v, n) in sorted(acc.items()): p(f"{k:66s} calls_per_pass={n // PASSES:5d} {c:26s} sum_per_pass={v / PASSES:20.1f}")
summ.close()
PY
  find "$OUT" -type f -size +2M -delete
  find "$OUT" -name "*.db" -delete
done
du -sh gpurun_out/prof_${TAG}_*
