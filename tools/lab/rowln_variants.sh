#!/usr/bin/env bash
# Lab: timing-only builds of the row-complete GEMM + LayerNorm kernel (ROWLN_LAB bit mask, csrc/gemm_rowln.hip) and their
# timing next to the product build in one process:
#   tools/lab/rowln_variants.sh "1 2 4 8 16 17 ..." && python tools/lab/rowln_variants.py <same list>
set -euo pipefail
for v in $1; do
  bash "$(dirname "$0")/build_variant.sh" rowln$v "-DROWLN_LAB=$v" gemm_rowln.hip
done
