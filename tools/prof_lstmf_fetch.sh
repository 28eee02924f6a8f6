#!/bin/bash
# lstm_f<40> HBM traffic by what-if variant (GPU box): FETCH_SIZE / WRITE_SIZE of one 6720-tile launch per library
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for n in base 1 2; do
  lib=$GRAFT_REPO_ROOT/voice-activity-detection-vad-onnx_amd/_exp/libvadx_d$n.so
  [ $n = base ] && lib=$GRAFT_REPO_ROOT/voice-activity-detection-vad-onnx_amd/libvadx.so
  for c in FETCH_SIZE WRITE_SIZE; do
    out=gpurun_out/lstmf_pmc/$n.$c; rm -rf $out; mkdir -p $out
    VADX_LIBRARY=$lib rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o p -- python3 tools/lstm_f_occ.py > $out/log.txt 2>&1
    python3 - $out $n $c <<'PY'
import csv, glob, sys
csv.field_size_limit(1 << 30)
out, n, c = sys.argv[1:4]
tot, cnt = 0.0, 0
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "lstm_f_kernel" in row["Kernel_Name"] and int(row["Grid_Size"]) == 6720 * 128:
            tot += float(row["Counter_Value"]); cnt += 1
print(f"variant {n}: {c} per 6720-tile launch = {tot / max(cnt, 1) / 1e6:.3f} M KiB over {cnt} rows")
PY
  done
done
