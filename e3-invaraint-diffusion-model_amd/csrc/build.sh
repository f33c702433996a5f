#!/usr/bin/env bash
# Build the C-ABI HIP library for gfx950 in-tree (the .so travels to the GPU box with gpurun).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libe3d_hip.so
SRCS="capi.hip gemm_f32.hip gemm_split.hip gemm_rowln.hip gemm_skinny.hip attn_relkey.hip attn_relkey_split.hip attn_relkey_coop.hip rowops.hip sampler.hip train_ops.hip attn_bwd.hip attn_bwd_split.hip attn_bwd_coop.hip nerf.hip dropout.hip optim.hip"
OBJS=""
pids=()
for s in $SRCS; do
  o="${s%.hip}.o"
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ e3d_common.h -nt "$o" ] || [ ../../include/e3d_hip.h -nt "$o" ]; then
    $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $(realpath $OUT)"
