#!/bin/bash
# compile csrc/dfsmn_cfb.hip to /tmp and report registers / spills / scratch traffic per barrier interval (development aid)
cd /root/repo/voice-activity-detection-vad-onnx_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-const-variable "$@" -c dfsmn_cfb.hip -o /tmp/cfb_v4.o -save-temps=obj 2>&1 | grep -E "error|warning" | head
S=/tmp/dfsmn_cfb-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|name):" $S | paste - - - - | sed 's/\s\+/ /g' | head -3
for k in 15cfb_back_kernel 16cfb_front_kernelILi20 16cfb_front_kernelILi40; do
  awk "/^_ZN4vadx9dfsmn_cfb$k/,/s_endpgm/" $S > /tmp/k.s; echo "== $k"
  grep -n "scratch_load\|scratch_store\|s_barrier\|v_mfma" /tmp/k.s | awk '{print $1,$2}' | awk '{if ($2=="v_mfma_f32_16x16x4_f32") {n++} else if ($2=="s_barrier") { printf("[%d mfma, %d sld, %d sst] | ", n, l, t); n=0; l=0; t=0 } else if ($2 ~ /scratch_load/) {l++} else {t++} } END {print ""}'
done
