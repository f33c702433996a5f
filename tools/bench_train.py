#!/usr/bin/env python3
"""Training-step timings on one GPU (BASELINE configs 2 and 4, per-GPU part): full-size models,
synthetic BioLiP-shaped batches, forward + loss + backward + grad-clip + AdamW.

    python tools/bench_train.py [structure|sequence] [--batch B] [--seq-len L] [--steps K]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402

DEV = "cuda:0"


def run(model_name="structure", batch=None, seq_len=128, steps=8, dropout=0.0, warmup=2, device=DEV, ddp=False, layers=None,
        seed=0, arithmetic=None, graph=None, trim=False):
    """One GPU's training step of BASELINE config 2 (structure, B=32) / config 4 (sequence, B=64): forward + loss +
    backward + gradient-norm clip + fused AdamW on synthetic BioLiP-shaped batches.  Returns a dict.
    ``ddp``: the step of ``training.fit`` under an initialised process group (BASELINE config 4: one rank per GPU,
    per-rank batch 64; ``trim``: the same batch on the frame of its longest ligand / pocket, ``training.trim_batch``)
    -- weights broadcast from rank 0, gradients as views of the all-reduce buckets, the buckets sent
    (RCCL; gloo in rehearsals) while the deferred weight-gradient launches of the later layers still run; timed between
    barriers, max over ranks by the caller."""
    DEV = device   # noqa: N806 (shadows the module default)
    L = seq_len
    layers = layers or (12 if model_name == "structure" else 6)
    B = batch or (32 if model_name == "structure" else 64)
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=layers,
             max_position_embeddings=L, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    enc, dec = BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True)
    torch.manual_seed(0)
    if model_name == "structure":
        from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
        from e3diff_amd.structure_model.dataset import noise_batch_on_device
        from e3diff_amd.structure_model.utils import CosineTables
        model = M(enc, dec, feature_names=list("abcdefgh"), loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4,
                  l2_lambda=0.1)
        tab = CosineTables(1000)
    else:
        from e3diff_amd.sequence_model.model import PeptideDiff as M
        model = M(enc, dec, feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(),
                  noise_schedule="cosine", timesteps=50, l2_lambda=0.1)
    model = model.train().to(DEV)
    optim = model.configure_optimizers()["optimizer"]
    params = [p for p in model.parameters() if p.requires_grad]
    pk = {k: v.to(DEV) for k, v in synthetic_pockets(B, L, seed=seed, with_ligand_seq=True).items() if torch.is_tensor(v)}
    frame = (L, L)
    if trim:
        frame = pkg.training.trimmed_frame(pk)
        if ddp:
            from e3diff_amd import sharding as _sh
            frame = _sh.max_over_ranks_host(frame)
        pk = pkg.training.trim_batch(pk, frame)
    averager = None
    if ddp:
        import torch.distributed as dist
        from e3diff_amd import sharding
        sharding.broadcast_parameters(model, src=0)
        averager = sharding.GradientAverager(model.parameters())

    # as training.fit does for a single process: the step replayed from a HIP graph (training.GraphedStep) after two eager
    # steps; ``graph=False`` / E3D_TRAIN_GRAPH=0: eager
    graph = (pkg.training.GRAPH_TRAIN if graph is None else graph) and isinstance(optim, pkg.optim.ClipAdamW)
    stepper = None
    if graph and not ddp:
        stepper = pkg.training.GraphedStep(model, optim, params, 1.0)
    elif graph and ddp and averager._active() and averager._hooked:
        # as training.fit does under a process group: forward + backward and clip + AdamW as two graph segments around the
        # eager all-reduce of the gradient buckets (training.GraphedDDPStep)
        stepper = pkg.training.GraphedDDPStep(model, optim, params, 1.0, averager)
    warmup = max(warmup, 4) if stepper is not None else warmup

    def step():
        if model_name == "structure":
            batch_ = dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab))
        else:
            batch_ = pk
        if stepper is not None:
            return stepper.step(batch_)
        loss = model.training_step(batch_)
        optim.zero_grad(set_to_none=True)
        if averager is not None:
            averager.prepare()
        if pkg.training.DEFER_WEIGHT_GRADS:        # as training.fit does: weight gradients grouped at the end of backward
            on_param = averager.mark_ready if (averager is not None and averager._active()) else None
            with pkg.autograd.deferred_weight_grads(on_param=on_param):
                loss.backward()
        else:
            loss.backward()
        if averager is not None:
            averager.average()
        pkg.training.clip_and_step(params, optim, 1.0)      # as training.fit does
        return loss

    with pkg.ops.arithmetic(arithmetic or pkg.training.TRAIN_ARITHMETIC):   # bf16x3 unless E3D_GEMM_MODE says otherwise
        mode = pkg.ops.GEMM_MODE
        for _ in range(warmup):
            loss = step()
        del loss
        torch.cuda.synchronize()
        if ddp and dist.is_initialized():
            dist.barrier()
        # host side alone: the time Python needs to ENQUEUE one step into an empty queue (the step is host-bound when this
        # approaches ms_per_step)
        h0 = time.perf_counter()
        step()
        host_ms = (time.perf_counter() - h0) * 1e3
        torch.cuda.synchronize()
        if ddp and dist.is_initialized():
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        if ddp and dist.is_initialized():
            dist.barrier()
        dt = (time.perf_counter() - t0) / steps
    if ddp and dist.is_initialized():
        from e3diff_amd import sharding
        dt = sharding.max_over_ranks(dt)
        w = torch.cat([p.detach().reshape(-1)[:64] for p in params]).double().sum().reshape(1)
        ws = [torch.zeros_like(w) for _ in range(dist.get_world_size())]
        if dist.get_backend() != "nccl":
            w = w.cpu()
            ws = [t.cpu() for t in ws]
        dist.all_gather(ws, w)
        assert all(float(t) == float(ws[0]) for t in ws), "ranks diverged"
    return {"model": model_name, "batch": B, "seq_len": L, "frame": list(frame), "layers": layers, "params_M": sum(p.numel() for p in params) / 1e6,
            "arithmetic": mode, "dropout": dropout, "graph_replay": bool(stepper is not None and stepper.graph is not None),
            "ms_per_step": dt * 1e3, "samples_per_s": B / dt, "host_enqueue_ms": host_ms,
            **({"ranks": dist.get_world_size(), "global_batch": B * dist.get_world_size(),
                "global_samples_per_s": B * dist.get_world_size() / dt, "backend": dist.get_backend(),
                "gradient_MB_per_step": sum(p.numel() for p in params) * 4 / 1e6,
                "buckets": len(averager.buckets)} if (ddp and dist.is_initialized()) else {}),
            "loss": float(loss.detach()), "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="structure", choices=["structure", "sequence"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--dropout", type=float, default=0.0, help="hidden and attention-probability dropout (reference: 0.1)")
    ap.add_argument("--eager", action="store_true", help="no HIP-graph replay of the step")
    ap.add_argument("--arithmetic", default=None, help="bf16x3 (default) | bf16x6 | bf16 = plain bf16 products, the reference's own training precision")
    ap.add_argument("--trim", action="store_true", help="the batch on the frame of its longest ligand / pocket (training.trim_batch)")
    args = ap.parse_args()
    r = run(args.model, args.batch, args.seq_len, args.steps, args.dropout, arithmetic=args.arithmetic,
            graph=False if args.eager else None, trim=args.trim)
    print(f"{r['model']} training step: B={r['batch']} L={r['seq_len']} frame={r['frame'][0]}x{r['frame'][1]} layers={r['layers']} params={r['params_M']:.1f}M "
          f"gemm_mode={r['arithmetic']} dropout={r['dropout']} graph={r['graph_replay']}: {r['ms_per_step']:.1f} ms/step = {r['samples_per_s']:.1f} samples/s "
          f"(loss {r['loss']:.4f}, peak mem {r['peak_mem_GiB']:.1f} GiB; host enqueue {r['host_enqueue_ms']:.1f} ms)", flush=True)


if __name__ == "__main__":
    main()
