#!/usr/bin/env bash
# Round-end evidence run on the GPU box: kernel-trace stats of the default bench line and the PMC
# (FETCH_SIZE / WRITE_SIZE, separate passes, no trace domains) traffic of the rel-key attention kernel.
# Outputs land in gpurun_out/prof_round/; copy the summaries into profiles/ afterwards.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $O/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o a -- python3 $R/tools/bench_kernels.py attn_pmc > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o a -- python3 $R/tools/bench_kernels.py attn_pmc > $O/pmc_write.log 2>&1
# the dominant GEMM kernel symbol over one whole bench step (every launch of it, as bench.py's `roofline` prices it)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_gfetch -o g -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only > $O/pmc_gfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_gwrite -o g -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only > $O/pmc_gwrite.log 2>&1
# the row-complete GEMM + LayerNorm kernel alone (6 launches at K = 768, then 6 at K = 1024; M = 65536)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_rfetch -o r -- python3 $R/tools/lab/rowln_one.py > $O/pmc_rfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_rwrite -o r -- python3 $R/tools/lab/rowln_one.py > $O/pmc_rwrite.log 2>&1
find $O -name "*.csv" | head -20
# the rel-key attention kernel at L = 64 / 128 / 256 (rocprofv3 evidence instead of HIP events only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_attn -o attn -- python3 $R/tools/bench_kernels.py attn_shapes > $O/attn_shapes.log 2>&1
# training steps (BASELINE configs 2 and 4, one GPU's share)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_structure -o t -- python3 $R/tools/bench_train.py structure --steps 5 > $O/train_structure.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train_sequence -o t -- python3 $R/tools/bench_train.py sequence --steps 5 > $O/train_sequence.log 2>&1
# ONE 64-residue pocket, 50 reverse steps (BASELINE configs[0] on the GPU): graph replay of the small-M kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -o s -- python3 $R/tools/bench_single.py > $O/single_pocket.log 2>&1
find $O -name "*stats.csv" | head -20
