#!/usr/bin/env python3
"""Lab: which parameters differ between two ranks (one GPU, gloo) after the data-parallel graph-segment step, run after run
in one process: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/lab/ddp_second_run_diff.py [runs]"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd import ops, sharding, training  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402
from e3diff_amd.sequence_model.model import PeptideDiff as M  # noqa: E402

rank, world, _ = sharding.init_distributed("gloo")
DEV = "cuda:0"
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = os.environ.get("MODE", "graph")
for run in range(runs):
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=2, max_position_embeddings=64,
             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("ACDEFGHIKLMNPQRSTVWY"),
              loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine", timesteps=50, l2_lambda=0.1).train().to(DEV)
    optim = model.configure_optimizers()["optimizer"]
    params = [p for p in model.parameters() if p.requires_grad]
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    pk = {k: v.to(DEV) for k, v in synthetic_pockets(4, 64, seed=rank, with_ligand_seq=True).items() if torch.is_tensor(v)}
    sharding.broadcast_parameters(model, src=0)
    avg = sharding.GradientAverager(model.parameters(), overlap=os.environ.get("OVERLAP", "1") == "1")
    stepper = training.GraphedDDPStep(model, optim, params, 1.0, avg, warmup=(2 if mode == "graph" else 10 ** 9))
    if rank == 0:
        print(f"   run {run}: stepper stream {stepper.stream.cuda_stream:#x}", flush=True)
    with ops.arithmetic("bf16x3"):
        for k in range(10):
            stepper.step(pk)
            if os.environ.get("PER_STEP") == "1":
                torch.cuda.synchronize()
                gs = torch.stack([(p.grad.detach().double().sum() if p.grad is not None else torch.zeros((), dtype=torch.double, device=DEV))
                                  for p in params]).cpu()
                both = [torch.zeros_like(gs) for _ in range(world)]
                dist.all_gather(both, gs)
                badg = [names[i] for i in range(len(names)) if float(both[0][i]) != float(both[1][i])]
                if rank == 0 and badg:
                    print(f"   run {run} step {k}: {len(badg)} gradients differ between the ranks after averaging: {badg[:8]}", flush=True)
    torch.cuda.synchronize()
    sums = torch.stack([p.detach().double().sum() for p in params]).cpu()
    both = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(both, sums)
    bad = [(names[i], float(both[0][i] - both[1][i])) for i in range(len(names)) if float(both[0][i]) != float(both[1][i])]
    if rank == 0:
        print(f"run {run} ({mode}, replaying={stepper.graph is not None}): {len(bad)} of {len(names)} parameters differ between the ranks; first: {bad[:6]}", flush=True)
    del model, optim, stepper, avg
    if os.environ.get("GC") == "1":
        import gc
        gc.collect()
        torch.cuda.empty_cache()
