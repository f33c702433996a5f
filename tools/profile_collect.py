#!/usr/bin/env python3
"""Turn the outputs of tools/profile_round.sh (gpurun_out/prof_round/) into the committed evidence under profiles/:
kernel-stats CSV, the bench line printed under rocprofv3, the PMC counter rows and traffic.json.

    python tools/profile_collect.py <tag>        # e.g. r01_v5
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "prof_round")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
B, L, H = 256, 256, 768


def counter_rows(path, needles):
    """rows of the kernels whose name contains one of ``needles`` (rocprofv3 prints demangled or mangled names
    depending on the run)"""
    return [r for r in csv.DictReader(open(path)) if any(n in r["Kernel_Name"] for n in needles)]


def mean(rows):
    v = [float(r["Counter_Value"]) for r in rows]
    return sum(v) / len(v), len(v)


def dump(rows, name):
    cols = ["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
            "Accum_VGPR_Count", "SGPR_Count"]
    with open(os.path.join(P, name), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore")
        w.writeheader()
        w.writerows(rows)


GEMM = ("gemm_split256p_kernel<0", "gemm_split256p_kernelILi0E")   # <ACT_NONE, element type>, demangled / mangled
ATTN = ("attn_coop_kernel<4, true", "attn_coop_kernelILi4ELb1E")   # 4-wave workgroups since round 4 (8-wave: E3D_ATTN_W=8)
af, aw = counter_rows(f"{O}/pmc_fetch/a_counter_collection.csv", ATTN), \
    counter_rows(f"{O}/pmc_write/a_counter_collection.csv", ATTN)
gf, gw = counter_rows(f"{O}/pmc_gfetch/g_counter_collection.csv", GEMM), counter_rows(f"{O}/pmc_gwrite/g_counter_collection.csv", GEMM)
(afm, _), (awm, _), (gfm, ng), (gwm, _) = mean(af), mean(aw), mean(gf), mean(gw)
traffic = {
    "source": f"profiles/{tag}_attn_coop_B256_L256_pmc_{{fetch,write}}.csv, profiles/{tag}_gemm_act0_bench_step_pmc_{{fetch,write}}.csv "
              f"(rocprofv3 --pmc passes of tools/profile_round.sh, collected by tools/profile_collect.py {tag})",
    f"attn_relkey_B{B}_L{L}_hbm_bytes_per_launch": (2 * afm + awm) * 1024,
    f"gemm_act0_B{B}_L{L}_hbm_bytes_per_launch": (2 * gfm + gwm) * 1024,
    "_detail": {
        "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes: FETCH_SIZE / WRITE_SIZE in KiB from separate rocprofv3 --pmc passes "
                   "(no trace domains); x2 = the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section)",
        "attn": {"kernel": "attn_coop_kernel<4 waves, rel-key, default arithmetic>, tools/bench_kernels.py attn_pmc", "FETCH_SIZE_KiB_raw_avg": afm,
                 "WRITE_SIZE_KiB_avg": awm, "algorithmic_bytes": (4 * L * H * 4 + (2 * L - 1) * 256 + 4 * L) * B},
        "gemm": {"kernel": f"{GEMM[0]}, ...>: every launch of one bench.py --headline-only step ({ng} dispatches incl. the untimed "
                           "trace pass)", "FETCH_SIZE_KiB_raw_avg": gfm, "WRITE_SIZE_KiB_avg": gwm},
        "command": "bash tools/profile_round.sh && python tools/profile_collect.py",
    },
}
# the row-complete GEMM + LayerNorm kernel (tools/lab/rowln_one.py: the first half of its launches are K = 768, the second K = 1024)
if os.path.exists(f"{O}/pmc_rfetch/r_counter_collection.csv") and os.path.exists(f"{O}/pmc_rwrite/r_counter_collection.csv"):
    rf, rw = counter_rows(f"{O}/pmc_rfetch/r_counter_collection.csv", ("gemm_rowln_kernel",)), counter_rows(f"{O}/pmc_rwrite/r_counter_collection.csv", ("gemm_rowln_kernel",))
    if rf and rw and len(rf) == len(rw) and len(rf) % 2 == 0:
        h = len(rf) // 2
        for K, lo, hi in ((768, 0, h), (1024, h, 2 * h)):
            f, w = mean(rf[lo:hi])[0], mean(rw[lo:hi])[0]
            traffic[f"gemm_rowln_M65536_K{K}_hbm_bytes_per_launch"] = (2 * f + w) * 1024
            traffic["_detail"].setdefault("rowln", {})[f"K{K}"] = {"FETCH_SIZE_KiB_raw_avg": f, "WRITE_SIZE_KiB_avg": w,
                                                                   "algorithmic_bytes": 4 * (65536 * K + 2 * 65536 * 768 + 768 * K + 3 * 768)}
        dump(rf, f"{tag}_gemm_rowln_M65536_pmc_fetch.csv")
        dump(rw, f"{tag}_gemm_rowln_M65536_pmc_write.csv")
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
dump(af, f"{tag}_attn_coop_B256_L256_pmc_fetch.csv")
dump(aw, f"{tag}_attn_coop_B256_L256_pmc_write.csv")
dump(gf, f"{tag}_gemm_act0_bench_step_pmc_fetch.csv")
dump(gw, f"{tag}_gemm_act0_bench_step_pmc_write.csv")
shutil.copy(f"{O}/stats/bench_kernel_stats.csv", os.path.join(P, f"{tag}_bench_B256_L256_kernel_stats.csv"))
for sub, name in (("stats_attn/attn_kernel_stats.csv", "attn_relkey_L64_L128_L256_kernel_stats.csv"),
                  ("stats_train_structure/t_kernel_stats.csv", "train_structure_B32_L128_kernel_stats.csv"),
                  ("stats_train_sequence/t_kernel_stats.csv", "train_sequence_B64_L128_kernel_stats.csv"),
                  ("stats_single/s_kernel_stats.csv", "single_pocket_L64_T50_kernel_stats.csv")):
    if os.path.exists(f"{O}/{sub}"):
        shutil.copy(f"{O}/{sub}", os.path.join(P, f"{tag}_{name}"))
for log in ("attn_shapes.log", "train_structure.log", "train_sequence.log", "single_pocket.log"):
    if os.path.exists(f"{O}/{log}"):
        shutil.copy(f"{O}/{log}", os.path.join(P, f"{tag}_{log}"))
line = [ln for ln in open(f"{O}/bench_under_rocprof.log") if ln.startswith("{")][-1]
open(os.path.join(P, f"{tag}_bench_line_under_rocprof.json"), "w").write(line)
d = json.loads(line)
rows = list(csv.DictReader(open(f"{O}/stats/bench_kernel_stats.csv")))
for r in rows[:6]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us {r['Percentage']:>6s} %")
print("bench under rocprof: roofline avg_launch_ms", d["roofline"]["avg_launch_ms"], "| attention", d["roofline_attention"]["avg_launch_ms"])
print("traffic per launch: attention", traffic[f"attn_relkey_B{B}_L{L}_hbm_bytes_per_launch"] / 1e6, "MB; GEMM",
      traffic[f"gemm_act0_B{B}_L{L}_hbm_bytes_per_launch"] / 1e6, "MB (algorithmic", d["roofline"]["algorithmic_bytes_per_launch"] / 1e6, "MB)")
