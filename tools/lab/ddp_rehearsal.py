#!/usr/bin/env python3
"""Two-rank rehearsal of the data-parallel training step on ONE GPU (both ranks on cuda:0, gloo for the
collectives): E3D_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1
tools/lab/ddp_rehearsal.py.  Checks that the overlapped gradient averaging gives every rank the mean of the
per-rank gradients and that the ranks' weights stay identical after optimizer steps."""
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
from e3diff_amd import sharding  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402
from e3diff_amd.structure_model.dataset import noise_batch_on_device  # noqa: E402
from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M  # noqa: E402
from e3diff_amd.structure_model.utils import CosineTables  # noqa: E402

rank, world, _ = sharding.init_distributed()
dev = "cuda:0"
c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=64,
         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(rank)      # different initial weights per rank: broadcast must fix that
model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
          loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.0).train().to(dev)
sharding.broadcast_parameters(model, src=0)
optim = torch.optim.AdamW(model.parameters(), lr=1e-3)
avg = sharding.GradientAverager(model.parameters(), bucket_bytes=1 << 20)
tab = CosineTables(100)
params = [p for p in model.parameters() if p.requires_grad]
def flat_grads():
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).cpu()


for step in range(3):
    pk = {k: v.to(dev) for k, v in synthetic_pockets(4, 64, seed=10 * step + rank).items() if torch.is_tensor(v)}

    def loss_fn():
        torch.manual_seed(100 + 10 * step + rank)          # same timestep / noise draw in both passes
        return model.training_step(dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab)))

    # reference: plain local backward, then an explicit mean over the ranks
    for p in params:
        p.grad = None
    loss_fn().backward()
    want = flat_grads()
    dist.all_reduce(want, op=dist.ReduceOp.SUM)
    want /= world
    # the product path: grads as bucket views, collectives launched from the autograd hooks
    optim.zero_grad(set_to_none=True)
    avg.prepare()
    loss = loss_fn()
    loss.backward()
    early = len(avg._handles)
    avg.average()
    got = flat_grads()
    err = ((got - want).abs().max() / want.abs().max()).item()
    assert err < 1e-5, err                                 # float atomics in a few reductions: last-bit noise only
    optim.step()
    print(f"rank {rank} step {step}: loss {float(loss.detach()):.4f}, averaged-gradient error {err:.1e}, "
          f"buckets launched during backward {early}/{len(avg.buckets)}", flush=True)
w = torch.cat([p.detach().reshape(-1) for p in params]).cpu()
ws = [torch.zeros_like(w) for _ in range(world)]
dist.all_gather(ws, w)
assert torch.equal(ws[0], ws[1]), "ranks diverged"
print(f"rank {rank}: weights identical across ranks after 3 steps", flush=True)
dist.barrier()
dist.destroy_process_group()
