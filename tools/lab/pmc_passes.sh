#!/usr/bin/env bash
# Lab: rocprofv3 --pmc passes (one counter group per run, no trace domains) over a small script; prints per-kernel sums.
#   tools/lab/pmc_passes.sh <out-dir> <script.py> "<group 1>" "<group 2>" ...
set -euo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; S=$R/$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- python3 $S > $O/g$i.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/g*/p_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "attn" in k or "gemm" in k:
            print(k, {c: round(v / n[(k, c)]) for c, v in d.items()}, "(avg per dispatch)")
PY
