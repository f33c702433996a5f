#!/usr/bin/env python3
"""Lab: find where a graph-replayed training step departs from the eager one.  Runs the soak of train_soak.py (structure,
dropout 0) under graph replay with a rolling snapshot of parameters + optimizer state; at the first step whose gradient norm
is not finite it restores the snapshot, repeats that very step EAGERLY on the same batch, and lists the parameters whose
gradients differ."""
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

DEV, L, B, steps = "cuda:0", 128, 32, 400
c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=12, max_position_embeddings=L,
         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
          loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1, learning_rate=1e-4).train().to(DEV)
tab = CosineTables(1000)
optim = model.configure_optimizers()["optimizer"]
params = [p for p in model.parameters() if p.requires_grad]
names = {id(p): n for n, p in model.named_parameters()}
stepper = training.GraphedStep(model, optim, params, 1.0)
pool = [{k: v.to(DEV) for k, v in synthetic_pockets(B, L, seed=s, with_ligand_seq=True).items() if torch.is_tensor(v)} for s in range(8)]
with ops.arithmetic(training.TRAIN_ARITHMETIC):
    for k in range(steps):
        pk = pool[k % len(pool)]
        batch = dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab))
        optim.param_groups[0]["lr"] = 1e-4 * min(1.0, (k + 1) / 50) * (1.0 - 0.5 * k / steps)
        snap = None
        if k >= 250:
            snap = ([p.detach().clone() for p in params], [(optim.state[p]["exp_avg"].clone(), optim.state[p]["exp_avg_sq"].clone()) for p in params])
        loss = stepper.step(batch)
        norm = float(optim.last_norm)
        if norm != norm or norm == float("inf"):
            print(f"step {k}: graph-replayed loss {float(loss):.5f}, gradient norm {norm}; timesteps {batch['timestep'].flatten().tolist()[:8]}...")
            tabx = optim._e3d_tab
            part = tabx["partial"].clone()
            badc = (~torch.isfinite(part)).nonzero().flatten().tolist()
            ct = tabx["chunk_tensor"].cpu().tolist()
            print(f"   non-finite per-chunk partial sums: {len(badc)} of {part.numel()}; tensors: {sorted({names[id(tabx['params'][ct[c]])] for c in badc})[:6]}")
            for c in badc[:3]:
                pp = tabx["params"][ct[c]]
                first = int(tabx["chunk_first"][c])
                seg = pp.grad.reshape(-1)[first:first + 8192]
                print(f"      chunk {c}: tensor {names[id(pp)]} elements [{first}, {first + seg.numel()}): .grad there finite={bool(torch.isfinite(seg).all())} max |g| {float(seg.abs().max()):.3e}; table ptr == .grad ptr: {int(tabx['gptr'][ct[c]]) == pp.grad.data_ptr()}")
            lib = pkg.hip.lib()
            nc2 = torch.zeros(2, device=DEV)
            pkg.hip.check(lib.e3d_grad_global_norm(tabx["gptr"].data_ptr(), tabx["numel"].data_ptr(), tabx["chunk_tensor"].data_ptr(),
                                                  tabx["chunk_first"].data_ptr(), tabx["n_chunks"], 1.0, tabx["partial"].data_ptr(), nc2.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "norm again")
            print(f"   the same norm launch repeated eagerly after the replay: {float(nc2[0])}")
            g_graph = [p.grad.detach().clone() for p in params]
            with torch.no_grad():
                for p, v in zip(params, snap[0]):
                    p.copy_(v)
            ops.invalidate_weight_caches()
            loss_e = model.training_step(batch)
            optim.zero_grad(set_to_none=True)
            with autograd.deferred_weight_grads():
                loss_e.backward()
            print(f"   eager loss on the same weights and batch: {float(loss_e):.5f}")
            rows = []
            for p, gg in zip(params, g_graph):
                ge = p.grad
                d = float((gg - ge).abs().max())
                rows.append((d / (float(ge.abs().max()) + 1e-30), float(gg.abs().max()), float(ge.abs().max()), names[id(p)]))
            bad = [r for r in rows if not (r[0] < 1e-2)]
            print(f"   parameters whose gradients differ by more than 1 %: {len(bad)} of {len(rows)}")
            order = {n: i for i, (n, _) in enumerate(model.named_parameters())}
            for r in sorted(bad, key=lambda r: -order[r[3]])[:12]:
                print(f"      {r[3]:60s} graph max |g| {r[1]:.3e}  eager {r[2]:.3e}")
            break
    else:
        print("no non-finite gradient norm in", steps, "steps")
