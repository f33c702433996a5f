#!/usr/bin/env python3
"""Generate the measurement table of DESIGN.md section 5 and the round's rows of profiles/README.md FROM the committed files
under profiles/ (three different figures for one JSON file could coexist when the cells were typed by
hand).  Rewrites the text between the ``<!-- BEGIN GENERATED <tag> ... -->`` / ``<!-- END GENERATED <tag> -->`` markers.

    python tools/profile_summary.py r04
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def jload(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    text = [ln for ln in open(path) if ln.lstrip().startswith("{")]
    return json.loads(text[-1]) if text else None


def stats(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return []
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    return [(r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / tot) for r in rows]


def find(rows, *needles):
    """(calls, avg_us, percent) summed over the symbols containing every needle (mangled or demangled names)"""
    hit = [r for r in rows if all(n in r[0] for n in needles)]
    calls = sum(r[1] for r in hit)
    avg = sum(r[1] * r[2] for r in hit) / calls if calls else float("nan")
    return calls, avg, sum(r[3] for r in hit)


def either(rows, *alts):
    for needles in alts:
        c = find(rows, *needles)
        if c[0]:
            return c
    return 0, float("nan"), 0.0


def log_lines(name, needle):
    path = os.path.join(P, name)
    return [ln.strip() for ln in open(path)] if os.path.exists(path) and needle is None else \
        [ln.strip() for ln in open(path) if needle in ln] if os.path.exists(path) else []


final = jload(f"{tag}_bench_line_final.json")
under = jload(f"{tag}_bench_line_under_rocprof.json")
traffic = json.load(open(os.path.join(P, "traffic.json")))
bench = stats(f"{tag}_bench_B256_L256_kernel_stats.csv")
attn = stats(f"{tag}_attn_relkey_L64_L128_L256_kernel_stats.csv")
tr_s = stats(f"{tag}_train_structure_B32_L128_kernel_stats.csv")
tr_q = stats(f"{tag}_train_sequence_B64_L128_kernel_stats.csv")
rccl = jload(f"{tag}_rccl_single_rank_step.json")

rows = []


def row(what, value, source):
    rows.append(f"| {what} | {value} | {source} |")


if final:
    src = f"`profiles/{tag}_bench_line_final.json` (`python bench.py`, steps {final['steps']}, warm-up {final['warmup']})"
    by = final.get("value_by_gemm_mode", {})
    row("`value` (f16x3, dense key sweep, encoder recomputed every step)",
        f"**{final['value']:.0f} pocket-steps/s** = {final['ms_per_step']:.1f} ms per step of 256 pockets"
        + (f" (timed first in the process; the same steps after the command's other legs, on a card that has been under full load for a minute: {final['ms_per_step_after_extras']:.1f} ms)" if final.get("ms_per_step_after_extras") else "")
        + ("; other arithmetics on the same line: " + ", ".join(f"{k} {v:.0f}" for k, v in by.items() if k != final.get('config', {}).get('gemm_mode', 'f16x3') and k != "f16x3") if len(by) > 1 else ""), src)
    row("`model_tflops` (SURVEY §8(d) \"structure full\" flops ÷ step time)", f"{final.get('model_tflops', float('nan')):.0f} TFLOP/s algorithmic", src)
    for key, label in (("value_padding_skip", "padding skip (bound-checked)"), ("value_encoder_cached", "+ encoder cached"), ("value_trimmed", "+ trimmed frame")):
        if final.get(key):
            row(f"product option, never in `value`: {label}", f"{final[key]:.0f} pocket-steps/s", src)
    r = final["roofline"]
    row("`roofline` (dominant symbol `gemm_split256p_kernel<ACT_NONE>`)",
        f"{r['launches_per_step']} launches per step × {r['avg_launch_ms'] * 1e3:.1f} µs (HIP events) = {r['achieved']:.0f} of {r['peak']:.0f} TFLOP/s = **{r['frac']:.3f}**; "
        f"traffic {r['traffic'] / 1e6:.0f} MB per launch vs {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic", src + ", `profiles/traffic.json`")
    ra = final["roofline_attention"]
    row("`roofline_attention` (rel-key, B = 256, L = 256)",
        f"{ra['launches_per_step']} launches per step × {ra['avg_launch_ms'] * 1e3:.1f} µs = {ra['achieved']:.0f} GB/s algorithmic = **{ra['frac']:.3f}** of 8 TB/s "
        f"({ra.get('algorithmic_TFLOPs', float('nan')):.0f} TFLOP/s); traffic {ra['traffic'] / 1e6:.0f} MB vs 839 MB algorithmic", src)
    rl = final.get("roofline_gemm_layernorm")
    if rl:
        row("`roofline_gemm_layernorm` (`gemm_rowln_kernel`: dense + bias + residual + LayerNorm)",
            f"{rl.get('launches_per_step', '?')} launches per step × {rl['avg_launch_ms'] * 1e3:.1f} µs = {rl['achieved']:.0f} TFLOP/s = {rl['frac']:.3f} of {rl['peak']:.0f}"
            + (f"; traffic {rl['traffic'] / 1e6:.0f} MB per launch vs {rl['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic" if rl.get("traffic") else ""), src)
    t = final.get("train")
    if t:
        def ms(k):
            v = t.get(k)
            return f"{v['ms_per_step']:.1f}" if isinstance(v, dict) else "?"
        row("`train` (bf16x3, dropout 0, graph replay): structure B = 32 / sequence B = 64, L = 128",
            " / ".join(ms(k) for k in t if isinstance(t[k], dict) and "eager" not in k) + " ms per step ("
            + ", ".join(k for k in t if isinstance(t[k], dict) and "eager" not in k) + ")", src)
    sp = final.get("single_pocket")
    if isinstance(sp, dict):
        seq = sp.get("sequence_stage") or {}
        row("`single_pocket` (config 1 workload on the GPU: B = 1, L = 64, T = 50, graph replay)",
            f"structure chain {sp['ms_per_chain']:.1f} ms = {sp['ms_per_step']:.2f} ms per reverse step"
            + (f"; sequence chain {seq['ms_per_chain']:.1f} ms = {seq['ms_per_step']:.2f} ms per step" if seq else ""), src)
    jt = final.get("joint")
    if isinstance(jt, dict) and "padded" in jt:
        row("`joint` (config 5, one GPU's 128 pockets × L = 128: structure T = 1000 → device hand-over → sequence T = 50)",
            f"{jt['padded']['total_s']:.2f} s padded, {jt['trimmed']['total_s']:.2f} s trimmed = {jt['trimmed']['pockets_per_s']:.1f} pockets/s per GPU", src)
    cb = final.get("cpu_baseline")
    if cb:
        row("`cpu_baseline` (oracle, kind = port)", f"{cb['value']:.1f} {cb['unit']} on {cb['cores']} threads ({cb['sample']}) → GPU ÷ CPU ≈ {final['value'] / cb['value']:.0f}×", src)
if under and bench:
    src = f"`profiles/{tag}_bench_B256_L256_kernel_stats.csv`, `{tag}_bench_line_under_rocprof.json` (`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only`)"
    g = either(bench, ("gemm_split256p_kernelILi0E",), ("gemm_split256p_kernel<0",))
    g1 = either(bench, ("gemm_split256p_kernelILi1E",), ("gemm_split256p_kernel<1",))
    rl = either(bench, ("gemm_rowln_kernel",))
    a = either(bench, ("attn_coop_kernelILi4ELb1E",), ("attn_coop_kernel<4, true",))
    ac = either(bench, ("attn_coop_kernelILi4ELb0E",), ("attn_coop_kernel<4, false",))
    ln = either(bench, ("residual_layernorm_kernel",))
    row("rocprofv3 averages of the same command vs the HIP-event figures it printed",
        f"dominant GEMM {g[1]:.1f} µs ({g[2]:.1f} % of kernel time) vs `roofline.avg_launch_ms` {under['roofline']['avg_launch_ms'] * 1e3:.1f}; "
        f"rel-key attention {a[1]:.1f} µs ({a[2]:.1f} %) vs {under['roofline_attention']['avg_launch_ms'] * 1e3:.1f}; "
        f"GELU GEMM {g1[1]:.1f} µs ({g1[2]:.1f} %); row-complete GEMM + LayerNorm {rl[1]:.1f} µs × {rl[0]} calls ({rl[2]:.1f} %); cross attention {ac[1]:.1f} µs ({ac[2]:.1f} %); "
        f"standalone `residual_layernorm_kernel`: {ln[0]} calls in the whole trace ({ln[2]:.2f} %)", src)
if attn:
    lines = log_lines(f"{tag}_attn_shapes.log", "attn relkey")
    row("rel-key attention at 65 536 tokens, all keys valid (`tools/bench_kernels.py attn_shapes` under rocprofv3)",
        "; ".join(re.sub(r"^attn relkey \S+ ", "", ln) for ln in lines), f"`profiles/{tag}_attn_shapes.log`, `{tag}_attn_relkey_L64_L128_L256_kernel_stats.csv`")
for name, st, label in ((f"{tag}_train_structure.log", tr_s, "structure"), (f"{tag}_train_sequence.log", tr_q, "sequence")):
    if st:
        tot_calls = sum(r[1] for r in st)
        top = sorted(st, key=lambda r: -r[3])[:4]
        lines = log_lines(name, "training step")
        row(f"training step under rocprofv3 ({label})",
            (re.sub(r"^.*graph=\w+: ", "", lines[-1]).split(" (")[0] if lines else "?") + " under the profiler; top symbols: "
            + "; ".join(f"`{re.sub(r'[^A-Za-z0-9_<>,: ]', '', r[0])[:48]}` {r[1]} × {r[2]:.1f} µs = {r[3]:.1f} %" for r in top),
            f"`profiles/{name}`, `{name.replace('.log', '')}_B{'32' if label == 'structure' else '64'}_L128_kernel_stats.csv`".replace(f"{tag}_train_{label}_B", f"{tag}_train_{label}_B"))
tf = [ln for ln in log_lines(f"{tag}_train_final.log", "training step")]
if tf:
    row("training step without the profiler (`tools/bench_train.py <model> --steps 20`, two runs each)",
        "; ".join(re.sub(r" training step:.*graph=\w+: ", ": ", ln).split(" = ")[0] for ln in tf), f"`profiles/{tag}_train_final.log`")
c3 = jload(f"{tag}_config3_full_chain.json")
if c3:
    row("BASELINE configs[2] end to end, once: the whole T = 1000 chain through `p_sample_loop` (product path: pocket encoder once per chain, bound-checked padding skip)",
        f"{c3['chain_s']:.1f} s for {c3['trajectory_shape'][0]} steps × {c3['trajectory_shape'][1]} pockets = {c3['ms_per_step']:.1f} ms per step = {c3['pocket_steps_per_s']:.0f} pocket-steps/s; "
        f"trajectory {c3['trajectory_GiB']:.2f} GiB on the device, copied out once in {c3['copy_out_s']:.2f} s; all finite: {c3['all_finite']}, inside [−π, π]: {c3['within_pi']}; peak memory {c3['peak_mem_GiB']:.1f} GiB",
        f"`profiles/{tag}_config3_full_chain.json` (`python tools/lab/config3_full_chain.py`)")
if rccl:
    g, e, s = rccl["rccl_one_rank_graph_segments"], rccl["rccl_one_rank_eager"], rccl["single_process_graph"]
    row("data-parallel step on RCCL, one-rank process group (the box has one GPU; `E3D_DDP_SINGLE_RANK=1`)",
        f"backend {rccl['backend']}, {g['buckets']} buckets, {g['gradient_MB_per_step']:.0f} MB of gradients through `all_reduce` per step: graph segments {g['ms_per_step']:.1f} ms "
        f"(host {g['host_enqueue_ms']:.1f} ms), eager {e['ms_per_step']:.1f} ms (host {e['host_enqueue_ms']:.1f} ms), no process group {s['ms_per_step']:.1f} ms; "
        f"losses agree to {max(rccl['loss_agreement'].values()):.1e}", f"`profiles/{tag}_rccl_single_rank_step.json` (`python tools/lab/rccl_single_rank_step.py`)")

table = "| quantity | value | source (file under `profiles/`, command) |\n|---|---|---|\n" + "\n".join(rows)
block = f"<!-- BEGIN GENERATED {tag} (tools/profile_summary.py {tag}) -->\n{table}\n<!-- END GENERATED {tag} -->"
pat = re.compile(rf"<!-- BEGIN GENERATED {tag}.*?<!-- END GENERATED {tag} -->", re.S)
for path in (os.path.join(ROOT, "DESIGN.md"), os.path.join(P, "README.md")):
    text = open(path).read()
    if pat.search(text):
        open(path, "w").write(pat.sub(lambda m: block, text))
        print("rewrote the generated block of", os.path.relpath(path, ROOT))
    else:
        print("no marker for", tag, "in", os.path.relpath(path, ROOT))
print(table)
