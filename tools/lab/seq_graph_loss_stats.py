#!/usr/bin/env python3
"""Lab: sequence-model training loss on ONE batch with frozen weights (lr = 0), eager steps against graph replays: the draws
of t and of the categorical noise come from torch's device generator inside the step; both modes must sample the same
distribution (means within sampling error)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd import ops, training  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402
from e3diff_amd.sequence_model.model import PeptideDiff  # noqa: E402

c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=64,
         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
batch = {k: v.to("cuda:0") for k, v in synthetic_pockets(16, 64, seed=3, with_ligand_seq=True).items() if torch.is_tensor(v)}
for graphed in (False, True, False, True):
    torch.manual_seed(0)
    model = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("ACDEFGHIKLMNPQRSTVWY"),
                        loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine", timesteps=50, l2_lambda=0.0, lr=0.0).train().to("cuda:0")
    optim = model.configure_optimizers()["optimizer"]
    params = [p for p in model.parameters() if p.requires_grad]
    stepper = training.GraphedStep(model, optim, params, 1.0, warmup=2 if graphed else 10 ** 9)
    with ops.arithmetic("bf16x3"):
        losses = torch.tensor([float(stepper.step(batch)) for _ in range(200)])
    print(f"graphed={graphed}: replaying={stepper.graph is not None}  mean {losses[5:].mean():.4f}  std {losses[5:].std():.4f}  min {losses.min():.3f} max {losses.max():.3f}"
          f"  first {[round(float(x), 3) for x in losses[:8]]}", flush=True)
