#!/usr/bin/env python3
"""Lab: soak of the graph-replayed training step at full model size with the reference's dropout (0.1) and a learning rate
that changes every step: N steps on a stream of different synthetic batches; every loss must be finite, the windowed mean
must go down, and replays must stay replays (no silent fallback)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd import ops, training  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402

DEV = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "structure"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
L, B = 128, (32 if name == "structure" else 64)
c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=12 if name == "structure" else 6,
         max_position_embeddings=L, hidden_dropout_prob=float(os.environ.get("E3D_SOAK_DROPOUT", "0.1")),
         attention_probs_dropout_prob=float(os.environ.get("E3D_SOAK_DROPOUT", "0.1")))
enc, dec = BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True)
torch.manual_seed(0)
if name == "structure":
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
    from e3diff_amd.structure_model.dataset import noise_batch_on_device
    from e3diff_amd.structure_model.utils import CosineTables
    model = M(enc, dec, feature_names=list("abcdefgh"), loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1,
              learning_rate=1e-4)
    tab = CosineTables(1000)
else:
    from e3diff_amd.sequence_model.model import PeptideDiff as M
    model = M(enc, dec, feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine",
              timesteps=50, l2_lambda=0.1, lr=1e-4)
model = model.train().to(DEV)
optim = model.configure_optimizers()["optimizer"]
params = [p for p in model.parameters() if p.requires_grad]
eager = os.environ.get("E3D_SOAK_EAGER") == "1"
stepper = training.GraphedStep(model, optim, params, 1.0, warmup=10 ** 9 if eager else 2)
trim = os.environ.get("E3D_SOAK_TRIM") == "1"      # batches on the frame of their longest ligand / pocket: several graphs
pool = [{k: v.to(DEV) for k, v in synthetic_pockets(B, L, seed=s, with_ligand_seq=True,
                                                    rec_range=(20, (40, 70, 100, None)[s % 4]) if trim else (20, None)).items()
         if torch.is_tensor(v)} for s in range(8)]
if trim:
    pool = [training.trim_batch(b) for b in pool]
    print("frames (ligand x pocket rows):", sorted({(b["ligand_angles"].shape[1], b["receptor_angles"].shape[1]) for b in pool}))
losses, norms = [], []
trace = [int(v) for v in os.environ.get("E3D_SOAK_TRACE", "").split(":")] if os.environ.get("E3D_SOAK_TRACE") else None
t0 = time.perf_counter()
with ops.arithmetic(training.TRAIN_ARITHMETIC):
    for k in range(steps):
        pk = pool[k % len(pool)]
        batch = dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab)) if name == "structure" else pk
        if os.environ.get("E3D_SOAK_CONST_LR") != "1":
            optim.param_groups[0]["lr"] = 1e-4 * min(1.0, (k + 1) / 50) * (1.0 - 0.5 * k / steps)
        if os.environ.get("E3D_SOAK_SYNC") == "1":
            torch.cuda.synchronize()
        out_loss = stepper.step(batch)
        if os.environ.get("E3D_SOAK_SYNC") == "1":
            torch.cuda.synchronize()
        if os.environ.get("E3D_SOAK_NOALLOC") == "1":      # no per-step allocations: results into preallocated rows
            if k == 0:
                loss_buf, norm_buf = torch.zeros(steps, device=DEV), torch.zeros(steps, device=DEV)
            loss_buf[k:k + 1].copy_(out_loss.detach().reshape(1))
            norm_buf[k:k + 1].copy_(optim.last_norm.detach().reshape(1))
            losses.append(loss_buf[k])
            norms.append(norm_buf[k])
        else:
            losses.append(out_loss.detach().clone())
            norms.append(optim.last_norm.detach().clone())
        if trace and trace[0] <= k < trace[1]:
            top = sorted(((float(p.grad.abs().max()), n) for n, p in model.named_parameters() if p.grad is not None), reverse=True)[:3]
            print(f"   step {k}: loss {float(losses[-1]):.4f} norm {float(norms[-1]):.3e} top grads {[(f'{v:.2e}', n) for v, n in top]}", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ls = torch.stack(losses).double().cpu()
assert bool(torch.isfinite(ls).all()), "non-finite loss"
w = max(10, steps // 8)
print(f"{name}: {steps} steps in {dt:.1f} s ({dt / steps * 1e3:.1f} ms/step incl. batch prep), replaying={stepper.graph is not None}, "
      f"graphs={len(stepper.graphs)}, eager steps per signature={sorted(stepper.seen.values())}, failed={stepper.failed!r}")
ns = torch.stack(norms).double().cpu()
bad = (~torch.isfinite(ns)).nonzero().flatten().tolist()
print("   first non-finite gradient norms at steps:", bad[:8], "of", len(bad))
print("   windowed mean grad norm:", [round(float(ns[i:i + w].mean()), 3) for i in range(0, steps - w + 1, w)])
print("   windowed mean loss:", [round(float(ls[i:i + w].mean()), 4) for i in range(0, steps - w + 1, w)])
assert eager or (stepper.graph is not None and stepper.failed is None)
assert float(ls[w:2 * w].mean()) < float(ls[:w].mean()), "loss did not go down"
print("   steps per parameter:", {int(st["step"]) for st in optim.state.values()}, " dropout epoch:", int(ops.dropout_epoch(DEV)))

if bad:
    names = {id(p): n for n, p in model.named_parameters()}
    for p in params:
        g_ = p.grad
        st = optim.state.get(p, {})
        flags = [("grad", g_), ("param", p.data), ("exp_avg", st.get("exp_avg")), ("exp_avg_sq", st.get("exp_avg_sq"))]
        msg = [f"{k}: {int((~torch.isfinite(t)).sum())} non-finite of {t.numel()}" for k, t in flags if t is not None and not bool(torch.isfinite(t).all())]
        if msg:
            print("   ", names[id(p)], tuple(p.shape), "; ".join(msg))

tab = optim._e3d_tab
if tab is not None:
    dev_ptrs = tab["gptr"].cpu().tolist()
    cur = [p.grad.data_ptr() if p.grad is not None else 0 for p in tab["params"]]
    print("   gptr table entries that differ from the parameters' current .grad:", sum(a != b for a, b in zip(dev_ptrs, cur)), "of", len(cur))
    tn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in tab["params"] if p.grad is not None))
    print("   norm from .grad tensors:", float(tn), " last_norm from the kernels:", float(optim.last_norm))
    worst = max(((float(p.grad.abs().max()), n) for n, p in model.named_parameters() if p.grad is not None))
    print("   largest |grad| element:", worst)
