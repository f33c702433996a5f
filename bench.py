#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of batched structure-model sampling on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2]): structure_model sampling, batch 256 x 256-residue synthetic
BioLiP-shaped pockets per GPU, full 12+12-layer 768-wide model, fp32, random-init weights.
One "step" = one reverse-diffusion step of the whole per-GPU batch: the full denoiser forward as
the reference computes it (pocket encoder included, nothing cached) + DDPM update + wrap, inputs
resident in HBM.  value = pockets * steps / second over all ranks ("pocket-steps/s"; weak scaling:
every rank owns its own pockets, no collective on the data path).  The sampler's real loop
(encoder + cross-K/V computed once per batch -- same results) is timed too and reported as
``value_encoder_cached``; it is NOT the headline because it skips work the reference does.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__  # noqa: E402

H, NH, INTER, LAYERS = 768, 12, 1024, 12
PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBPS = 8000.0


def attn_flops(B, L):      # 6 L^2 H per (item, layer), all heads
    return 6.0 * L * L * H * B


def attn_bytes(B, L):      # Q,K,V read + O written + E + mask
    return (4 * L * H * 4 + (2 * L - 1) * 64 * 4 + 4 * L) * B


def structure_flops_per_pocket(L):
    """SURVEY 8(d) formulae, "structure full" column (ligand and receptor padded to L)."""
    enc = 8 * L * H * H + 4 * L * H * INTER + 6 * L * L * H
    dec = 16 * L * H * H + 4 * L * H * INTER + 6 * L * L * H + 4 * L * L * H
    se_rec = 14 * L * H * H + 8 * L * H * H + 6 * L * L * H + 16 * L * H * H
    se_t = 14 * 1 * H * H + 8 * L * H * H + 6 * L * L * H + 16 * L * H * H
    head = 2 * L * H * H + 2 * L * H * 8
    return LAYERS * enc + LAYERS * dec + se_rec + se_t + head


def build_model(L, device):
    pkg = __graft_entry__.load_package()
    pkg.hip.lib()
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
    common = dict(hidden_size=H, num_attention_heads=NH, intermediate_size=INTER, num_hidden_layers=LAYERS,
                  max_position_embeddings=L)
    torch.manual_seed(0)
    model = ConditionalBertForDiffusionBase(BertConfig(**common),
                                            BertConfig(**common, is_decoder=True, add_cross_attention=True), 8)
    # adaLN_modulation[0] is zero-initialised by the reference; give it weights so the gated
    # branches are live in the benchmark (random-init of the same architecture)
    with torch.no_grad():
        for se in (model.receptor_emb, model.timestep_emb):
            torch.nn.init.normal_(se.adaLN_modulation[0].weight, std=0.02)
    return model.eval().to(device), pkg


def host_cores():
    """CPUs this process may actually use: the smaller of its affinity mask and its cgroup CPU
    quota (the GPU box exposes 256 hardware threads but grants a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(L, seed):
    """Oracle (CPU restatement incl. the rel-key term) timed on the host cores: one full
    structure-model reverse step (forward + update + wrap) on a bounded sample of the workload."""
    from helpers import seeded_state_dict, synthetic_pockets
    from oracle import structure as ostr
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
    B, warm, timed = 8, 3, 10   # 3 warm-up + >= 10 timed steps; ~10 s of CPU work on 16 cores
    common = dict(hidden_size=H, num_attention_heads=NH, intermediate_size=INTER, num_hidden_layers=LAYERS,
                  max_position_embeddings=L)
    m = ConditionalBertForDiffusionBase(BertConfig(**common),
                                        BertConfig(**common, is_decoder=True, add_cross_attention=True), 8)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed)
    del m
    cfg = {"num_heads": NH, "max_pos": L}
    pk = synthetic_pockets(B, L, seed=seed)
    x = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8))
    betas = ostr.cosine_beta_schedule(1000)
    fn = lambda t, xx, lm, rs, ra, rm: ostr.forward(sd, cfg, t, xx, lm, rs, ra, rm)  # noqa: E731
    cores = host_cores()
    torch.set_num_threads(cores)
    times = []
    with torch.no_grad():
        for i in range(warm + timed):
            t0 = time.perf_counter()
            x = ostr.modulo_with_wrapped_range(ostr.p_sample(
                fn, pk["ligand_attn_mask"], x, pk["receptor_seq"], pk["receptor_attn_mask"],
                pk["receptor_angles"], torch.full((B,), 999 - i), betas))
            times.append(time.perf_counter() - t0)
            print(f"[cpu_baseline] step {i}: {times[-1]:.2f} s on {cores} threads", file=sys.stderr, flush=True)
    med = sorted(times[warm:])[len(times[warm:]) // 2]
    out = {"value": B / med, "unit": "pocket-steps/s", "cores": cores, "kind": "port",
           "sample": f"oracle (CPU restatement of the reference incl. relative_key), structure model 12+12 layers, "
                     f"B={B} x L={L} pockets, {warm} warm-up + {timed} timed reverse steps, median"}
    # BASELINE configs[0] end to end: ONE 64-residue pocket (ligand 12), T = 50, the whole
    # p_sample_loop as structure_model/sample.py:101-144 runs it (encoder recomputed every step, as the reference does)
    L1, T1 = 64, 50
    common = dict(hidden_size=H, num_attention_heads=NH, intermediate_size=INTER, num_hidden_layers=LAYERS,
                  max_position_embeddings=L1)
    m = ConditionalBertForDiffusionBase(BertConfig(**common),
                                        BertConfig(**common, is_decoder=True, add_cross_attention=True), 8)
    sd1 = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed)
    del m
    cfg1 = {"num_heads": NH, "max_pos": L1}
    pk1 = synthetic_pockets(1, L1, seed=seed, lig_range=(12, 12), rec_range=(64, 64))
    fn1 = lambda t, xx, lm, rs, ra, rm: ostr.forward(sd1, cfg1, t, xx, lm, rs, ra, rm)  # noqa: E731
    betas1 = ostr.cosine_beta_schedule(T1)
    x1 = ostr.modulo_with_wrapped_range(torch.randn(1, L1, 8))
    with torch.no_grad():
        for rep in range(2):          # first chain = warm-up
            t0 = time.perf_counter()
            ostr.p_sample_loop(fn1, pk1["ligand_attn_mask"], x1, pk1["receptor_seq"], pk1["receptor_attn_mask"],
                               pk1["receptor_angles"], T1, betas1)
            chain_s = time.perf_counter() - t0
            print(f"[cpu_baseline] config 1 chain {rep}: {chain_s:.2f} s on {cores} threads", file=sys.stderr, flush=True)
    out["config1"] = {"value": T1 / chain_s, "unit": "pocket-steps/s", "chain_s": chain_s, "cores": cores, "kind": "port",
                      "sample": f"the same oracle, B=1 x L={L1} (ligand 12, pocket 64), T={T1} reverse steps end to end "
                                "(second of two chains)"}
    return out


def launch_workers(n):
    """Self-contained multi-GPU launch: start ``n`` fresh copies of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, as torchrun would set them), one per GPU.  Runs BEFORE anything in this
    process touches the GPU and never re-execs it.  Rank 0 prints the JSON line (inherited stdout); the
    parent's exit code is the first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code and not rc:   # a dead rank would leave the others in the rendezvous / a collective forever
                    rc = code
                    for q in live:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


def plumbing_rehearsal(args, rank, world):
    """E3D_BENCH_REHEARSAL=cpu: the control path of a multi-rank run with no GPU work at all."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    dt = 0.001 * (1 + rank)        # rank-dependent "elapsed": the reduction must return the slowest rank's
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "denoising-steps/sec (batched pocket graphs)", "rehearsal": "cpu plumbing only",
                          "value": args.batch * world * args.steps / dt, "unit": "pocket-steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "max_elapsed_s": dt, "scaling": "weak"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="pockets per GPU")
    ap.add_argument("--seq-len", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the extra training-step timings")
    ap.add_argument("--no-joint-leg", action="store_true", help="skip the joint structure -> sequence chain (config 5 share)")
    ap.add_argument("--train-ddp-batch", type=int, default=64, help="per-rank batch of the train_ddp leg (N > 1)")
    ap.add_argument("--train-ddp-seq-len", type=int, default=128)
    ap.add_argument("--train-ddp-layers", type=int, default=6)
    ap.add_argument("--gemm-mode", default=os.environ.get("E3D_GEMM_MODE", "f16x3"),
                    choices=["f32", "bf16x3", "bf16x6", "f16x3"],
                    help="GEMM arithmetic of the headline value (see DESIGN.md section 3)")
    ap.add_argument("--only-default-mode", action="store_true", help="skip timing the other GEMM modes")
    ap.add_argument("--headline-only", action="store_true",
                    help="profile runs: only the headline loop (dense key sweep), no extras -- every attention "
                         "launch in a rocprofv3 trace of this command is then the launch `roofline` describes")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without an external launcher: this process never touches the GPU, it only
        # starts one fresh worker per GPU (no exec of an initialised process) and relays rank 0's JSON line
        sys.exit(launch_workers(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU)")
    rehearsal = os.environ.get("E3D_BENCH_REHEARSAL", "")
    if rehearsal == "cpu":
        # launcher / rendezvous / barrier / max-over-ranks / single-JSON-line plumbing without any GPU
        # (tests/test_sharding_cpu.py); never a measurement
        return plumbing_rehearsal(args, rank, world)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # E3D_BENCH_REHEARSAL=1: the multi-rank control path on a ONE-GPU box (all ranks on cuda:0, gloo for the
        # barrier / max-over-ranks) -- a plumbing check, never a measurement
        if rehearsal == "1":
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    B, L = args.batch, args.seq_len
    model, pkg = build_model(L, device)
    from helpers import synthetic_pockets
    from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
    from e3diff_amd.structure_model import sample as S

    pk = {k: v.to(device) for k, v in synthetic_pockets(B, L, seed=1000 + rank).items() if torch.is_tensor(v)}
    tab = CosineTables(1000)
    gen = torch.Generator(device=device).manual_seed(rank)
    x = modulo_with_wrapped_range(torch.randn(B, L, 8, device=device, generator=gen)).contiguous()
    nxt = torch.empty_like(x)

    def full_step(i, x, out):
        """What the reference does per reverse step: whole model + update + wrap."""
        return S._reverse_step(model, pk["ligand_attn_mask"], x, pk["receptor_seq"], pk["receptor_attn_mask"],
                               pk["receptor_angles"], i, tab, None, None, out, wrap=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, k, t_hi=999):
        nonlocal x, nxt
        barrier()
        t0 = time.perf_counter()
        for j in range(k):
            y = step_fn(t_hi - j, x, nxt)
            x, nxt = y, x
        barrier()
        return time.perf_counter() - t0

    by_mode = {}
    lib = pkg.hip.lib()
    with torch.no_grad():
        # product default: the attention key sweep stops after the last valid key's tile (bit-identical
        # results).  Timed first as an extra; the headline below runs the DENSE sweep over all padded keys,
        # i.e. every flop the reference does.
        pkg.ops.set_gemm_mode(args.gemm_mode)
        pkg.ops.set_attn_mode(args.gemm_mode)
        # THE HEADLINE FIRST (round 4): W untimed + K timed steps of the dense sweep, before any of the extras below touches
        # the card.  The extras take about a minute of full load, and a card that has been busy that long holds a 3-5 %
        # lower clock (DVFS); the same K steps are timed AGAIN after them and reported beside the headline as
        # ``ms_per_step_after_extras`` (what a long chain sustains), never as ``value``.
        lib.e3d_attn_skip_padded_tiles(0)
        timed(full_step, args.warmup)
        elapsed = timed(full_step, args.steps)
        by_mode[args.gemm_mode] = elapsed / args.steps
        lib.e3d_attn_skip_padded_tiles(1)
        elapsed_skip = float("nan")
        if not args.headline_only:
            timed(full_step, 1)
            elapsed_skip = timed(full_step, max(1, args.steps // 2)) / max(1, args.steps // 2)
        lib.e3d_attn_skip_padded_tiles(0)
        # the other GEMM arithmetic modes, for transparency (same work, same kernels otherwise)
        if not args.only_default_mode and not args.headline_only:
            for mode in pkg.ops.GEMM_MODES:
                if mode != args.gemm_mode:
                    pkg.ops.set_gemm_mode(mode)
                    pkg.ops.set_attn_mode(mode)
                    timed(full_step, 1)
                    by_mode[mode] = timed(full_step, max(1, args.steps // 2)) / max(1, args.steps // 2)
        pkg.ops.set_gemm_mode(args.gemm_mode)
        pkg.ops.set_attn_mode(args.gemm_mode)
        # the sampler's actual loop: receptor encoded once, padding skip on (product defaults)
        lib.e3d_attn_skip_padded_tiles(1)
        cache = model.encode_receptor(pk["receptor_seq"], pk["receptor_angles"], pk["receptor_attn_mask"])

        def cached_step(i, x, out):
            return S._reverse_step(model, pk["ligand_attn_mask"], x, None, None, None, i, tab, None, cache, out, True)

        elapsed_cached = float("nan")
        if not args.headline_only:
            timed(cached_step, 1)
            elapsed_cached = timed(cached_step, args.steps)

        # ... and with the frame trimmed to the longest ligand / pocket of the batch (p_sample_loop(trim_padding=True):
        # valid positions unchanged, the decoder runs on 32 of the 256 ligand rows for BioLiP-shaped batches)
        elapsed_trimmed = float("nan")
        if not args.headline_only:
            Ll, Lr = S.trimmed_length(pk["ligand_attn_mask"]), S.trimmed_length(pk["receptor_attn_mask"])
            cache_t = model.encode_receptor(pk["receptor_seq"][:, :Lr].contiguous(), pk["receptor_angles"][:, :Lr].contiguous(),
                                            pk["receptor_attn_mask"][:, :Lr].contiguous())
            mask_t = pk["ligand_attn_mask"][:, :Ll].contiguous()
            xa, xb = x[:, :Ll].contiguous(), torch.empty_like(x[:, :Ll].contiguous())
            for k_steps in (1, args.steps):
                barrier()
                t0 = time.perf_counter()
                for j in range(k_steps):
                    y = S._reverse_step(model, mask_t, xa, None, None, None, 999 - j, tab, None, cache_t, xb, True)
                    xa, xb = y, xa
                barrier()
                elapsed_trimmed = time.perf_counter() - t0
            del cache_t

        # the headline loop once more, on the card as the extras left it (hot)
        lib.e3d_attn_skip_padded_tiles(0)
        elapsed_after = float("nan")
        if not args.headline_only:
            timed(full_step, 1)
            elapsed_after = timed(full_step, args.steps)
        # per-launch durations of the named kernels (HIP events on the launch stream), one dense step
        pkg.ops.TRACE = []
        full_step(500, x, nxt)
        torch.cuda.synchronize()
        trace, pkg.ops.TRACE = pkg.ops.TRACE, None
        lib.e3d_attn_skip_padded_tiles(1)
    del cache

    modes = sorted(by_mode)
    if dist is not None:
        t = torch.tensor([elapsed, elapsed_cached, elapsed_skip, elapsed_trimmed, elapsed_after] + [by_mode[m] for m in modes],
                         device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_cached, elapsed_skip, elapsed_trimmed, elapsed_after, *rest = t.tolist()
        by_mode = dict(zip(modes, rest))

    def avg_ms(name):
        d = [a.elapsed_time(b) for n, a, b, _ in trace if n == name]
        return sum(d) / len(d), len(d)

    attn_ms, n_attn = avg_ms("attn_relkey")
    gemm_ms = sum(a.elapsed_time(b) for n, a, b, _ in trace if n == "gemm")
    gemm_flops = sum(2.0 * m[0] * m[1] * m[2] for n, _, _, m in trace if n == "gemm")
    by_shape = {}
    for n, ev0, ev1, m in trace:
        if n == "gemm":
            by_shape.setdefault(tuple(int(v) for v in m[:3]), []).append(ev0.elapsed_time(ev1))
    shapes = sorted(({"M": k[0], "N": k[1], "K": k[2], "launches_per_step": len(v), "avg_launch_ms": sum(v) / len(v),
                      "TFLOPs": 2.0 * k[0] * k[1] * k[2] / (sum(v) / len(v) * 1e-3) / 1e12} for k, v in by_shape.items()),
                    key=lambda d: -d["avg_launch_ms"] * d["launches_per_step"])

    # the dominant kernel SYMBOL: the plain-epilogue (bias, no activation) launches of the large-M GEMM -- in the
    # default mode gemm_split256p_kernel<0>, whose per-kernel average in a rocprofv3 --stats summary of this command
    # is directly comparable with avg_launch_ms below
    dom = [(ev0.elapsed_time(ev1), m) for n, ev0, ev1, m in trace
           if n == "gemm" and (len(m) < 4 or m[3] == 0) and m[0] % 256 == 0 and m[1] % 256 == 0
           and (m[0] // 256) * (m[1] // 256) >= 256]
    dom_ms = sum(t for t, _ in dom) / max(1, len(dom)) or float("nan")   # (no large-M launch at rehearsal sizes)
    dom_flops = sum(2.0 * m[0] * m[1] * m[2] for _, m in dom) / max(1, len(dom))
    dom_bytes = sum(4.0 * (m[0] * m[2] + m[1] * m[2] + m[0] * m[1] + m[1]) for _, m in dom) / max(1, len(dom))

    # the row-complete GEMM + bias + residual + LayerNorm launches (csrc/gemm_rowln.hip; BertSelfOutput / BertOutput)
    rowln = [(ev0.elapsed_time(ev1), m) for n, ev0, ev1, m in trace if n == "gemm_layernorm"]
    rowln_ms = sum(t for t, _ in rowln) / max(1, len(rowln)) or float("nan")
    rowln_flops = sum(2.0 * m[0] * m[1] * m[2] for _, m in rowln) / max(1, len(rowln))
    rowln_bytes = sum(4.0 * (m[0] * m[2] + m[1] * m[2] + 2 * m[0] * m[1] + 3 * m[1]) for _, m in rowln) / max(1, len(rowln))

    if rank == 0:
        traffic = gemm_traffic = traffic_source = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            tdict = json.load(open(tj))
            traffic = tdict.get(f"attn_relkey_B{B}_L{L}_hbm_bytes_per_launch")
            gemm_traffic = tdict.get(f"gemm_act0_B{B}_L{L}_hbm_bytes_per_launch")
            traffic_source = tdict.get("source")
        a_tf = attn_flops(B, L) / (attn_ms * 1e-3) / 1e12
        gemm_peak = {"f32": PEAK_F32_MATRIX_TFLOPS, "bf16x3": 2500.0 / 3, "bf16x6": 2500.0 / 6, "f16x3": 2500.0 / 3}[args.gemm_mode]
        a_gb = attn_bytes(B, L) / (attn_ms * 1e-3) / 1e9
        out = {
            "metric": "denoising-steps/sec (batched pocket graphs)",
            "value": B * world * args.steps / elapsed,
            "unit": "pocket-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_after_extras": None if (args.headline_only or elapsed_after != elapsed_after) else 1e3 * elapsed_after / args.steps,
            "ms_per_step_note": "ms_per_step / value: the W + K steps run FIRST in the process; ms_per_step_after_extras: the same K steps "
                                "timed again after the other legs of this command (~1 min of full load: the card then holds a lower clock)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 storage and accumulation, bf16x3 products",
                      "bf16x6": "f32 storage and accumulation, bf16x6 products (fp32 grade)",
                      "f16x3": "f32 storage and accumulation, f16x3 products (fp32 grade)"}[args.gemm_mode],
            "data": "synthetic",
            "gemm_mode": {"f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                          "bf16x3": "fp32 operands split into 2 bf16 terms, 3 cross products on the bf16 MFMA, fp32 accumulate",
                          "bf16x6": "fp32 operands split into 3 bf16 terms, 6 cross products on the bf16 MFMA, fp32 accumulate "
                                    "(fp32-grade)",
                          "f16x3": "fp32 operands split into 2 fp16 terms (11 + 11 bits), 3 cross products on the fp16 MFMA, fp32 "
                                   "accumulate (fp32-grade: 3.8e-6 end to end at L=256 vs 2.6e-6 for the exact fp32 MFMA path, both against "
                                   "the fp64 oracle; weights pre-scaled by an exact power of two so that both terms are normal "
                                   "fp16 numbers, activations must lie inside the fp16 range -- beyond it the outputs are inf/NaN)"}[args.gemm_mode],
            "value_by_gemm_mode": {m: B * world / by_mode[m] for m in modes},
            "config": {"workload": f"structure_model sampling step, {B} x {L}-residue pockets per GPU, "
                                   "12+12 layers x 768, T=1000 schedule, encoder recomputed every step (as the reference)",
                       "pockets_per_gpu": B, "seq_len": L, "parallelism": f"pocket-sharded x{world}, no collective"},
            "model_tflops": structure_flops_per_pocket(L) * B * world * args.steps / elapsed / 1e12,
            "value_encoder_cached": None if args.headline_only else B * world * args.steps / elapsed_cached,
            "value_padding_skip": None if args.headline_only else B * world / elapsed_skip,
            "value_trimmed": None if args.headline_only else B * world * args.steps / elapsed_trimmed,
            "value_notes": "value = dense attention sweep over all padded keys + encoder recomputed every step "
                           "(every flop the reference does); value_padding_skip = product default (key sweep stops "
                           "after the last valid key, bit-identical results); value_encoder_cached = the sampler's "
                           "real loop (pocket encoder + cross K/V once per batch), with the padding skip; value_trimmed = the same loop "
                           "on the frame trimmed to the batch's longest ligand / pocket (p_sample_loop(trim_padding=True), what "
                           "structure_model/sample.py's sample() runs: valid positions unchanged)",
            # the binding roofline of the rel-key attention kernel: whichever of MFMA time (algorithmic flops /
            # peak) and HBM time (algorithmic bytes / 8 TB/s) is larger; frac = that time / measured time
            "roofline_attention": (lambda t_mfma, t_hbm: {
                "kernel": ("attn_fwd_kernel<relkey> (e3d_relkey_attn_fwd)" if args.gemm_mode == "f32" else
                           f"attn_coop_kernel<4 waves, relkey> for bf16x3 / f16x3, attn_fwd_split_kernel / attn_fwd_kernel otherwise (e3d_relkey_attn_fwd_split, {args.gemm_mode})"),
                "bound": "hbm" if t_hbm > t_mfma else "mfma",
                "achieved": a_gb if t_hbm > t_mfma else a_tf,
                "peak": PEAK_HBM_GBPS if t_hbm > t_mfma else gemm_peak,
                "unit": "GB/s" if t_hbm > t_mfma else "TFLOP/s",
                "frac": max(t_hbm, t_mfma) / attn_ms, "traffic": traffic, "traffic_source": traffic_source,
                "peak_note": "MFMA peak: fp32 157.3 TFLOP/s for f32; bf16 / fp16 dense 2500 / cross products for the split modes",
                "avg_launch_ms": attn_ms, "launches_per_step": n_attn, "dense_key_sweep": True,
                "algorithmic_TFLOPs": a_tf, "mfma_frac": a_tf / gemm_peak,
                "hbm_algorithmic_GBps": a_gb, "hbm_frac": a_gb / PEAK_HBM_GBPS})(
                    attn_flops(B, L) / (gemm_peak * 1e12) * 1e3, attn_bytes(B, L) / (PEAK_HBM_GBPS * 1e9) * 1e3),
            # the dominant kernel (~56 % of the step's kernel time): all its launches of one step.  flops per launch =
            # 2MNK, bytes per launch = 4(MK + NK + MN + N), averaged over those launches; peak = the
            # dense bf16 MFMA rate / cross products per fp32 product
            "roofline": {"kernel": ("gemm_split256p_kernel<ACT_NONE> (e3d_gemm_bias_act_f32_split), " if args.gemm_mode in ("bf16x3", "f16x3")
                                    else f"large-M GEMM kernel of mode {args.gemm_mode}, ") +
                                   f"all {len(dom)} launches of one step (M={B * L}; N, K in gemm_shapes)",
                         "bound": "mfma", "achieved": dom_flops / (dom_ms * 1e-3) / 1e12, "peak": gemm_peak, "unit": "TFLOP/s",
                         "frac": dom_flops / (dom_ms * 1e-3) / 1e12 / gemm_peak, "traffic": gemm_traffic,
                         "traffic_source": traffic_source,
                         "traffic_note": None if not (gemm_traffic and dom_bytes) else
                         f"{gemm_traffic / dom_bytes:.2f}x the algorithmic bytes: the A row block is re-fetched by the N tiles "
                         "of its row (served by the memory-side cache; priced at ~7 % of the launch by the fixed-operand "
                         "ablation, profiles/README.md)",
                         "avg_launch_ms": dom_ms, "launches_per_step": len(dom),
                         "algorithmic_flops_per_launch": dom_flops, "algorithmic_bytes_per_launch": dom_bytes,
                         "peak_note": "fp32 MFMA 157.3 for f32; bf16 dense 2500 / terms for the split modes; traffic = HBM "
                                      "bytes per launch (average over the same launches) from the PMC passes in "
                                      "profiles/traffic.json"},
            # BertSelfOutput / BertOutput as one launch each (round 4): 2MNK flops; A + weight planes + residual in, out
            # written once -- the GEMM + LayerNorm pair it replaces moves one more [M, 768] tensor out and two more in
            "roofline_gemm_layernorm": {
                "kernel": "gemm_rowln_kernel (e3d_gemm_residual_layernorm_f32_split): dense -> + bias + residual -> LayerNorm, "
                          f"all {len(rowln)} launches of one step",
                "bound": "mfma", "achieved": rowln_flops / (rowln_ms * 1e-3) / 1e12, "peak": gemm_peak, "unit": "TFLOP/s",
                "frac": rowln_flops / (rowln_ms * 1e-3) / 1e12 / gemm_peak, "avg_launch_ms": rowln_ms,
                "launches_per_step": len(rowln), "algorithmic_flops_per_launch": rowln_flops,
                "algorithmic_bytes_per_launch": rowln_bytes,
                "hbm_algorithmic_GBps": rowln_bytes / (rowln_ms * 1e-3) / 1e9 if rowln else None,
                # HBM-side bytes per launch from the PMC passes of tools/profile_round.sh (per K, M = 65536), averaged over the
                # step's own mix of K; null when a launch has a shape the passes did not cover
                "traffic": (lambda per: (sum(per) / len(per)) if per and all(v is not None for v in per) else None)(
                    [tdict.get(f"gemm_rowln_M{int(m[0])}_K{int(m[2])}_hbm_bytes_per_launch") if os.path.exists(tj) else None
                     for _, m in rowln]),
                "traffic_source": traffic_source,
                "replaces": "gemm_split256p_kernel<ACT_NONE> + residual_layernorm_kernel (E3D_GEMM_ROWLN=0 runs that pair)"},
            "gemm_shapes": shapes,
            "roofline_gemm": {"kernel": f"GEMM ({args.gemm_mode}), all launches of one step", "bound": "mfma",
                              "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12,
                              "peak": gemm_peak, "unit": "TFLOP/s (algorithmic 2MNK)",
                              "frac": gemm_flops / (gemm_ms * 1e-3) / 1e12 / gemm_peak,
                              "peak_note": "fp32 MFMA 157.3 for f32; bf16 dense 2500 / terms for the split modes",
                              "ms_per_step": gemm_ms},
        }
        if world == 1 and not args.headline_only and not args.no_train_leg:
            # extra keys (never the headline): one training step of BASELINE configs 2 and 4, one GPU's share
            del model
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_train
            out["train"] = {"structure_B32_L128": bench_train.run("structure", steps=10),
                            "sequence_B64_L128": bench_train.run("sequence", steps=10),
                            "structure_B32_L128_eager": bench_train.run("structure", steps=5, graph=False),
                            "structure_B32_L128_trimmed": bench_train.run("structure", steps=10, trim=True),
                            "sequence_B64_L128_trimmed": bench_train.run("sequence", steps=10, trim=True),
                            "note": "forward + loss + backward + grad-norm clip + AdamW, 12+12 / 6 layers x 768, synthetic "
                                    "batches, dropout 0 (the reference's 0.1 costs +0.7 ms); the step as training.fit runs it "
                                    "for one process: captured into a HIP graph after two eager steps and replayed "
                                    "(graph_replay; host_enqueue_ms = Python time per step); *_eager = the same step "
                                    "launch by launch; *_trimmed = the same batches on the frame of their longest ligand / "
                                    "pocket (training.fit(trim_padding=True), opt-in: same loss and gradients, `frame` rows)"}
            import bench_single
            # BASELINE configs[0] on the GPU: ONE 64-residue pocket, 50 reverse steps (latency, not throughput)
            out["single_pocket"] = bench_single.run(seq_len=64, batch=1, steps=50)
            out["single_pocket"]["sequence_stage"] = bench_single.run_sequence(seq_len=64, batch=1, steps=50)
            if not args.no_joint_leg:
                # BASELINE configs[4], one GPU's share: 128 pockets x L=128, structure T=1000 -> hand-over on the device ->
                # sequence T=50 (sequence_model/sample_by_generated_angles.py:196-278); the reference's padded frames,
                # then the frames trimmed to the longest ligand / pocket (valid positions unchanged)
                import bench_joint
                out["joint"] = {"padded": bench_joint.run(128, 128, 1000, 50, trim=False, device=str(device)),
                                "trimmed": bench_joint.run(128, 128, 1000, 50, trim=True, device=str(device)),
                                "note": "config 5 = 1024 pockets over 8 GPUs: this is one rank's 128 (pocket-sharded, no "
                                        "collective until the final gather)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, seed=0)
    if world > 1 and not args.headline_only and not args.no_train_leg:
        # extra key for N > 1 (never the headline): BASELINE config 4 -- the sequence model's data-parallel training
        # step, per-rank batch 64, gradients averaged over the ranks (RCCL all-reduce of the bucket views, overlapped
        # with the deferred weight-gradient launches).  Every rank runs it; rank 0 reports the max-over-ranks time.
        del model
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_train
        try:
            ddp = bench_train.run("sequence", batch=args.train_ddp_batch, seq_len=args.train_ddp_seq_len, steps=5, warmup=2,
                                  device=str(device), ddp=True, layers=args.train_ddp_layers, seed=rank)
            ddp["note"] = ("sequence model (reference sequence_model/train_model.py:93-104 under DDP), forward + loss + "
                           "backward + gradient all-reduce / world + clip + AdamW; ms_per_step = max over ranks")
        except Exception as e:   # noqa: BLE001 -- an extra key must never cost the headline line (the same code path on
            ddp = {"error": f"{type(e).__name__}: {e}"}   # every rank: a Python error is raised by all of them alike)
        if rank == 0:
            out["train_ddp"] = ddp
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
