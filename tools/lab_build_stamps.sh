#!/usr/bin/env bash
# Lab build: the library with ${LAB_DEFS:--DE3D_STAMPS} (phase time stamps in the 256x256 GEMM) -> lab_build/libe3d_stamps.so
set -euo pipefail
cd "$(dirname "$0")/../e3-invaraint-diffusion-model_amd/csrc"
mkdir -p ../../lab_build/obj
objs=""
for s in capi gemm_f32 gemm_split attn_relkey attn_relkey_split attn_relkey_coop rowops sampler train_ops attn_bwd attn_bwd_split nerf dropout; do
  if [ $s = gemm_split ] || [ $s = attn_relkey_coop ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 ${LAB_DEFS:--DE3D_STAMPS} -c $s.hip -o ../../lab_build/obj/$s.o
    objs="$objs ../../lab_build/obj/$s.o"
  else
    objs="$objs $s.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../lab_build/libe3d_stamps.so
echo built lab_build/libe3d_stamps.so
