"""Lab: where do the small torch launches of a training step (fills, adds, cats, copies) come from?  torch.profiler with
stacks over one structure training step; prints aten ops by calling frame inside this repository."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__
pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets
from e3diff_amd.bert import BertConfig
from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
from e3diff_amd.structure_model.dataset import noise_batch_on_device
from e3diff_amd.structure_model.utils import CosineTables
DEV = "cuda:0"
L, B = 128, 32
c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=12, max_position_embeddings=L,
         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
          loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1).train().to(DEV)
optim = model.configure_optimizers()["optimizer"]
params = [p for p in model.parameters() if p.requires_grad]
pk = {k: v.to(DEV) for k, v in synthetic_pockets(B, L, seed=0).items() if torch.is_tensor(v)}
tab = CosineTables(1000)

def step():
    loss = model.training_step(dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab)))
    optim.zero_grad(set_to_none=True)
    with pkg.autograd.deferred_weight_grads():
        loss.backward()
    pkg.training.clip_and_step(params, optim, 1.0)

with pkg.ops.arithmetic("bf16x3"):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::cat", "aten::copy_",
        "aten::full", "aten::mul", "aten::empty_like")
by = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        frame = next((f for f in ev.stack if "e3-invaraint" in f or "/tools/" in f or "torch/optim" in f or "torch/nn/utils" in f
                      or "autograd/function" in f), ev.stack[0] if ev.stack else "?")
        by[(ev.name, frame.split("/")[-1][:90])] += 1
for (name, frame), n in by.most_common(45):
    print(f"{n:5d} {name:18s} {frame}")
