#!/usr/bin/env python3
"""Lab: per-parameter difference between the gradients a graph-replayed training step leaves in .grad and the gradients an
eager step computes from the same weights and batch (small structure model, dropout 0), at an early step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd import autograd, ops, training  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402
from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M  # noqa: E402
from e3diff_amd.structure_model.dataset import noise_batch_on_device  # noqa: E402
from e3diff_amd.structure_model.utils import CosineTables  # noqa: E402

DEV, L, B = "cuda:0", 128, 32
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
at = int(sys.argv[2]) if len(sys.argv) > 2 else 8
c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=layers, max_position_embeddings=L,
         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
          loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1, learning_rate=1e-4).train().to(DEV)
with torch.no_grad():
    for se in (model.receptor_emb, model.timestep_emb):
        torch.nn.init.normal_(se.adaLN_modulation[0].weight, std=0.02)
tab = CosineTables(1000)
optim = model.configure_optimizers()["optimizer"]
params = [p for p in model.parameters() if p.requires_grad]
names = {id(p): n for n, p in model.named_parameters()}
stepper = training.GraphedStep(model, optim, params, 1.0)
pool = [{k: v.to(DEV) for k, v in synthetic_pockets(B, L, seed=s).items() if torch.is_tensor(v)} for s in range(4)]
with ops.arithmetic(training.TRAIN_ARITHMETIC):
    for k in range(at + 1):
        pk = pool[k % len(pool)]
        batch = dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab))
        if k == at:
            snap = [p.detach().clone() for p in params]
        stepper.step(batch)
    assert stepper.graph is not None
    g_graph = [p.grad.detach().clone() for p in params]
    with torch.no_grad():
        for p, v in zip(params, snap):
            p.copy_(v)
    ops.invalidate_weight_caches()
    loss_e = model.training_step(batch)
    optim.zero_grad(set_to_none=True)
    with autograd.deferred_weight_grads():
        loss_e.backward()
rows = []
for p, gg in zip(params, g_graph):
    ge = p.grad
    rows.append((float((gg - ge).abs().max()) / (float(ge.abs().max()) + 1e-30), names[id(p)], tuple(p.shape)))
rows.sort(reverse=True)
print(f"step {at}: parameters with graph-vs-eager gradient difference > 1e-4 (relative to max |g|): {sum(r[0] > 1e-4 for r in rows)} of {len(rows)}")
for r in rows[:14]:
    print(f"   {r[0]:.3e}  {r[1]}  {r[2]}")
