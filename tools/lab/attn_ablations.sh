#!/usr/bin/env bash
# Lab: TIMING-ONLY ablation builds of the cooperative attention kernel (-DE3D_ATTN_ABL=<mask>, see attn_relkey_coop.hip) into
# lab_build/libe3d_attn_abl<mask>.so from the archived source with the switches (tools/lab/archive/attn_relkey_coop_r04_ablation_switches.hip.txt;
# the product kernel has none).   tools/lab/attn_ablations.sh 1 2 4 8 ...
set -euo pipefail
cd "$(dirname "$0")/../../e3-invaraint-diffusion-model_amd/csrc"
mkdir -p ../../lab_build/obj
cp ../../tools/lab/archive/attn_relkey_coop_r04_ablation_switches.hip.txt ./_attn_abl_tmp.hip
trap "rm -f _attn_abl_tmp.hip" EXIT
OBJS=$(ls *.o | grep -v "^attn_relkey_coop.o$" | tr '\n' ' ')
pids=()
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -DE3D_ATTN_ABL=$m -c _attn_abl_tmp.hip -o ../../lab_build/obj/abl${m}_attn_relkey_coop.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../lab_build/obj/abl${m}_attn_relkey_coop.o -o ../../lab_build/libe3d_attn_abl${m}.so ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls ../../lab_build/libe3d_attn_abl*.so
