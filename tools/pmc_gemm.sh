#!/usr/bin/env bash
# PMC passes over the dominant GEMM launch (one counter group per run; no trace domains with --pmc).
set -euo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_gemm/g$i -o p -- python3 $R/tools/bench_kernels.py ${1:-gemm_pmc} > $R/gpurun_out/pmc_gemm_$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_gemm/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_split" in r["Kernel_Name"] or "attn_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} launches {len(v)}  mean {sum(v)/len(v):.4g}")
PY
